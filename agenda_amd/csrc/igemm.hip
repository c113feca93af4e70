// Implicit-GEMM convolution / linear kernel for gfx950 (CDNA4), bf16 in, fp32 accumulate.
//
// Replaces on the reference path (all reached through diffusers from
// reference data_generation/data_generation.py:59): ResnetBlock2D 3x3 convs, Down/Upsample2D
// convs, Transformer2DModel proj_in/out 1x1 convs, every nn.Linear (to_q/k/v/out, GEGLU FF),
// and the VAE decoder convs (SURVEY.md §8a rows U1,U4,U6,U7,V1).
//
// Design (MI355X-first, not a cuDNN/cuBLAS translation):
//  * activations live NHWC bf16, so a conv tap is a contiguous 128-B channel run per pixel and a
//    transformer token row IS a pixel row -- no NCHW<->NHWC transposes anywhere;
//  * A (im2col rows) and B (weights [N][K], K = tap-major) tiles go global->LDS by LDS-DMA
//    (buffer_load_dwordx4 ... lds): wave-uniform buffer descriptors, a per-lane 32-bit byte offset
//    that is fixed for a whole (tap, source) segment = the im2col gather, and the K advance in an
//    SGPR soffset -- zero VALU per load in steady state; padding pixels get an out-of-range offset
//    and the hardware range check writes zeros; STAGES-deep ring, one barrier per 64-deep K step;
//  * LDS rows are 128 B with the 16-B chunk index XOR (row&7): conflict-free ds_read_b128 for the
//    16x16x32 MFMA operand fetch; the XOR is applied on the SOURCE address (LDS-DMA writes
//    lane-linear);
//  * skip-connection concat, nearest-2x upsample and stride-2 are folded into the gather;
//  * MFMA operand roles are swapped (D = W_tile . X_tile^T) and the weight rows permuted inside a wave's N range, so a
//    lane's accumulators are consecutive output channels of one pixel: the epilogue (bias / time-embedding row add /
//    residual / GEGLU / SiLU, one rounding to bf16) runs straight from registers and stores whole 16-B row chunks --
//    no LDS staging, no barrier (igemm_epilogue.h);
//  * block->tile map is XCD-aware (tiles sharing an A panel share an L2).
#include "kernels.h"
#include <cstdio>
#include <cstdlib>
#ifdef AGD_STAMPS   // `make stamps` (libagenda_hip_stamps.so = the experiments build + these marks; the experiments build proper carries none, so its kernels are the product's)
// in-kernel time stamps of one wave (tools/kb_*_trace.py; AGD_IGEMM_CFG bit 10 = dbg & 64): the chosen wave of workgroup (g_smap_ts_wg, z = 0) stores s_memtime at every mark;
// a kernel declares `const bool ts_on = ...; int ts_n = 0;` and finishes with g_smap_ts[1023] = ts_n
__device__ unsigned long long g_smap_ts[1024];
__device__ int g_smap_ts_wg;
#define AGD_TS(k) do { if (ts_on && ts_n < 1000) { g_smap_ts[ts_n++] = ((unsigned long long)(k) << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull); } } while (0)
#else
#define AGD_TS(k) do { } while (0)
#endif
#include "igemm_epilogue.h"
#include "igemm_halo.h"
#include "igemm8p.h"
#include "igemm_wreg.h"
#include "igemm_smap.h"
#include "igemm_pc.h"
#include "igemm_pch.h"
#include <type_traits>
#include <cstdlib>
#define CK0(expr) do { if ((expr) != 0) return -1; } while (0)

template <int N> AGD_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

#define OOB_OFF 0x80000000u      // >= num_records (0x7FFFFFF0): buffer range check returns zeros

// KG = 2 (plain 1x1 launches of one source, no split-K; launches of <= 256 tiles, i.e. ONE workgroup per CU): the workgroup is two groups of
// WM x WN waves, group g runs K steps g, g + 2, ... through its own STAGES-slot ring into its own accumulators; group 1 hands its
// accumulators to group 0 through LDS before the epilogue.  One wave per SIMD is a chain of barrier -> fragment reads -> MFMAs per K step with
// nothing to overlap it (tools/kb_m512.py: the launch takes as long with every load dropped); two independent chains per SIMD overlap.
template <int BM, int BN, int WM, int WN, int KS, int STAGES, int GEGLU, int SPLITK, int KG = 1>
__global__ __launch_bounds__(WM * WN * 64 * KG, 2) void igemm_kernel(const IgemmP p) {
  constexpr int NT = WM * WN * 64;                   // threads of one K group
  constexpr int NW = WM * WN;
  static_assert(KG == 1 || (KG == 2 && KS == 1 && !GEGLU && !SPLITK), "K groups: plain 1x1 launches only");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int A_IT = (BM * 8 + NT - 1) / NT, B_IT = (BN * 8 + NT - 1) / NT;   // DMA instructions per wave (last may be partial)
  constexpr int A_Q = BM / 8, B_Q = BN / 8;          // 8-row DMA groups per tile
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for thread count");
  static_assert(STAGES == 2 || ((BM * 8) % NT == 0 && (BN * 8) % NT == 0), "counted vmcnt needs equal DMA counts per wave");
  constexpr int LPS = A_IT + B_IT;                   // LDS-DMA instructions per stage per wave
  static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 2) * LPS < 64, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef AGD_EXPERIMENTS
  if (p.dbg & 32) return;                          // timing experiment (AGD_IGEMM_CFG=512): dispatch cost of this grid only
#endif
  const int tid = threadIdx.x % NT, lane = tid & 63;
  const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  char* const ring = smem + kg * (STAGES * (BM + BN) * 128);
  // Phase stagger (short-K launches with several tiles per CU slot): the two workgroups that share a CU start in
  // lockstep and stay there -- both in the LDS/MFMA-bound main loop, then both in the store/VALU-bound epilogue.  The
  // second occupant of each CU (blocks 256..511 of the first dispatch wave) starts `stagger` x 1024 cycles late, so one
  // workgroup's epilogue runs beside the other's main loop for the rest of the launch.  Speed only: any placement is correct.
#ifdef AGD_EXPERIMENTS   // measured: no effect at any delay; kept for the experiments library only
  if (p.stagger > 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x >> 8) == 1) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
#endif
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  // consecutive logical ids share an XCD's L2 (xcd_remap): walk the tiles so that they share the LARGER operand panel --
  // A-major (same im2col rows, all N tiles) for the big feature maps, W-major (same weight rows, all M tiles) where the
  // weights outweigh the activations (16x16 / 8x8 maps): otherwise every XCD streams the whole weight matrix from the
  // fabric (rocprofv3 FETCH_SIZE: 211 MB per launch for a 29 MB weight matrix)
  int tn, tm;
  tile_of(bid, (p.M + BM - 1) / BM, tiles_n, p.wmajor, p.xb_m, p.xb_n, tm, tn);
  const int bz = blockIdx.y;
  const int m0 = tm * BM, n0 = tn * BN;

  // wave-uniform buffer descriptors (kernel args + blockIdx only)
  const bf16_t* base0 = p.src0 + bz * p.sA0;
  const bf16_t* base1 = p.src1 ? p.src1 + bz * p.sA1 : p.src0;
  const bf16_t* baseW = p.W + bz * p.sW + (p.w_per_image ? (long long)(m0 / (p.Hout * p.Wout)) * p.sW : 0);   // per-image matrices: GroupNorm folded into W
#ifdef AGD_EXPERIMENTS   // timing experiments only (p.dbg): zero-record descriptors drop the loads but keep the instruction stream
  const unsigned nrecA = (p.dbg & 1) ? 0u : 0x7FFFFFF0u, nrecB = (p.dbg & 2) ? 0u : 0x7FFFFFF0u;
#else
  constexpr unsigned nrecA = 0x7FFFFFF0u, nrecB = 0x7FFFFFF0u;
#endif

  // ---- per-thread gather state ---------------------------------------------------------
  const int lrow = lane >> 3;                       // row within the 8-row DMA group
  const int lchunk = (lane & 7) ^ lrow;             // logical 16-B chunk this lane fetches (swizzle on source)
  const int HWo = p.Hout * p.Wout;
  const int ush = (p.up == 2) ? 1 : 0;
  const int Hup = p.Hin << ush, Wup = p.Win << ush;
  int a_b[A_IT], a_y[A_IT], a_x[A_IT];
  unsigned a_rowok = 0;
  const bool lin = KS == 1 && p.stride == 1 && p.up == 1 && p.pad == 0 && p.Hin == p.Hout && p.Win == p.Wout;
  // (phase convs of an upsampling conv, IgemmP::ups4: this tile's phase (a, b) = n0 / Cout pads (1 - a) rows above and (1 - b) columns left of its 2x2 window)
  const int ph4 = (KS == 2 && p.ups4) ? n0 / p.ups4 : 0;
  const int pad_y = (KS == 2 && p.ups4) ? 1 - (ph4 >> 1) : p.pad, pad_x = (KS == 2 && p.ups4) ? 1 - (ph4 & 1) : p.pad;
  const bool small_m = p.M < (1 << 24);
  const float inv_hwo = 1.0f / (float)HWo, inv_wo = 1.0f / (float)p.Wout;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + (i * NW + wid) * 8 + lrow;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    if (lin) { a_b[i] = 0; a_y[i] = 0; a_x[i] = mm; }      // 1x1, stride 1: the im2col row IS the pixel index
    else {
      int b, oy;
      if (small_m) { b = fast_udiv(mm, HWo, inv_hwo); const int rem_ = mm - b * HWo; oy = fast_udiv(rem_, p.Wout, inv_wo); }
      else { b = mm / HWo; oy = (mm - b * HWo) / p.Wout; }
      const int ox = mm - b * HWo - oy * p.Wout;
      a_b[i] = b; a_y[i] = oy * p.stride - pad_y; a_x[i] = ox * p.stride - pad_x;
    }
    a_rowok |= (ok ? 1u : 0u) << i;
  }
  // B rows are read by the MFMA in permuted order (tile j, fragment row rho = 4q' + r' <-> wave-local row q'*4NI + 4j + r',
  // see igemm_epilogue.h), so the conflict-free swizzle key of a row is rho & 7 = (row & 3) | ((q' & 1) << 2)
  unsigned bvoff[B_IT];                              // fixed for the whole kernel
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = (i * NW + wid) * 8 + lrow;       // tile-local weight row this lane fetches
    const int n = n0 + row;
    const int qp = (row % WTN) / (4 * NI);
    const int key = (row & 3) | ((qp & 1) << 2);
    bvoff[i] = (n < p.N) ? (unsigned)(((long long)n * p.K + ((lane & 7) ^ key) * 8) * 2) : OOB_OFF;
  }
  unsigned avoff[A_IT];                              // fixed within a (tap, source) segment
  unsigned asoff = 0, bsoff = 0;                     // SGPR byte offsets: K advance
  int seg_left = 0, tap = 0, cursrc = 0;

  // K-steps of this block: [ks0, ks0 + nk)  (split-K: grid.z slices the K range)
  const int nk_total = p.K >> 6;
  int ks0 = kg, nk = KG == 1 ? nk_total : (nk_total - kg + KG - 1) / KG;      // K groups: steps kg, kg + KG, ...
  if constexpr (SPLITK) {
    constexpr int Q = KS * KS;                       // 3x3: slice on whole channel chunks (9 taps each)
    const int per = ((nk_total / Q + (int)gridDim.z - 1) / (int)gridDim.z) * Q;
    ks0 = (int)blockIdx.z * per;
    nk = nk_total - ks0 < per ? nk_total - ks0 : per;
    if (nk < 0) nk = 0;
  }

  auto set_segment = [&](int tap_, int src_, int off_steps) {
    tap = tap_; cursrc = src_;
    const int kh = (KS > 1) ? tap / KS : 0, kw = (KS > 1) ? tap - kh * KS : 0;
    const int Cs = cursrc ? p.C1 : p.C0;
    seg_left = KG == 1 ? (Cs >> 6) - off_steps : (1 << 30);     // K groups: one source, one tap (launcher) -- the segment never ends
    asoff = (unsigned)off_steps * 128u;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = a_y[i] + kh, ix = a_x[i] + kw;
      const bool ok = ((a_rowok >> i) & 1) && (lin || ((unsigned)iy < (unsigned)Hup && (unsigned)ix < (unsigned)Wup));
      const int pix = lin ? a_x[i] : (a_b[i] * p.Hin + (iy >> ush)) * p.Win + (ix >> ush);
      avoff[i] = ok ? (unsigned)(((long long)pix * Cs + lchunk * 8) * 2) : OOB_OFF;
    }
  };
  auto new_segment = [&]() {
    if (cursrc == 0 && p.C1 > 0) set_segment(tap, 1, 0); else set_segment(tap + 1, 0, 0);
  };
  {
    const int spt = (p.C0 + p.C1) >> 6;              // k-steps per tap
    const int t0 = ks0 / spt, r0 = ks0 - t0 * spt;
    if (r0 < (p.C0 >> 6)) set_segment(t0, 0, r0); else set_segment(t0, 1, r0 - (p.C0 >> 6));
    bsoff = (unsigned)ks0 * 128u;
  }

  // Cold-weight warm-up (W-major launches: the weight matrix is the big operand and arrives from HBM -- the UNet's 1.7 GB of weights never
  // stay in the 256 MB Infinity Cache).  The launch's first ~512 workgroups (co-resident) stream the contiguous slice of W that THEIR
  // XCD's tiles will read (xcd_remap gives every XCD a contiguous range of logical ids, W-major = a contiguous row range) through the
  // XCD's L2 at full memory-level parallelism, instead of every tile meeting its rows cold one 16 KB K step at a time.  The pieces land
  // in a scratch KiB per wave behind the ring; speed only (any block placement is correct).
  if (p.warm == 2) {
    // A-major launches with a large weight matrix (every XCD reads all of W): the first workgroups stream W once, 1/nb each, so the
    // other XCDs and later tiles find it in the Infinity Cache instead of HBM
    const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int tot = gridDim.x * gridDim.y * gridDim.z, nb = tot < 512 ? tot : 512;
    if (lin < nb) {
      const long long pieces = ((long long)p.N * p.K * 2) >> 10;
      const long long p0 = pieces * lin / nb, p1 = pieces * (lin + 1) / nb;
      char* wl = smem + KG * STAGES * STAGE + (kg * NW + wid) * 1024;
      for (long long pc = p0 + kg * NW + wid; pc < p1; pc += NW * KG) bufdma16(baseW, wl, (unsigned)(pc * 1024 + lane * 16), 0u);
    }
  } else if (p.warm && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x >> 3) < 64) {
    const int nwg = gridDim.x, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int lbase = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8, lcnt = q8 + (x < r8 ? 1 : 0);
    if (lcnt > 0) {
      const int tiles_m = (p.M + BM - 1) / BM;
      const int tn0 = lbase / tiles_m, tn1 = (lbase + lcnt - 1) / tiles_m;
      const int r0 = tn0 * BN, r1 = (tn1 + 1) * BN < p.N ? (tn1 + 1) * BN : p.N;
      const long long b0 = (long long)r0 * p.K * 2, pieces = ((long long)(r1 - r0) * p.K * 2) >> 10;
      const long long p0 = pieces * j / 64, p1 = pieces * (j + 1) / 64;
      char* wl = smem + KG * STAGES * STAGE + (kg * NW + wid) * 1024;
      for (long long pc = p0 + kg * NW + wid; pc < p1; pc += NW * KG) bufdma16(baseW, wl, (unsigned)(b0 + pc * 1024 + lane * 16), 0u);
    }
  }

  // prologue stage fill; live == false issues the same instructions through a zero-record descriptor (dropped):
  // keeps the per-wave vmcnt arithmetic of the deeper rings uniform when the K range is shorter than the ring
  auto issue = [&](int slot, bool live) {
#ifdef AGD_EXPERIMENTS
    if (p.dbg & 4) return;                      // timing experiment: no DMA instructions at all
#endif
    if (live && seg_left == 0) new_segment();
    char* sA = ring + slot * STAGE;
    char* sB = sA + A_BYTES;
    // make the scalar operands provably wave-uniform (else hipcc wraps every load in a waterfall loop)
    const unsigned aso = __builtin_amdgcn_readfirstlane(asoff), bso = __builtin_amdgcn_readfirstlane(bsoff);
    const unsigned nrA = live ? nrecA : 0u, nrB = live ? nrecB : 0u;
    if (__builtin_amdgcn_readfirstlane(cursrc)) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (i * NW + wid < A_Q) bufdma16(base1, sA + (i * NW + wid) * 1024, avoff[i], aso, nrA);
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (i * NW + wid < A_Q) bufdma16(base0, sA + (i * NW + wid) * 1024, avoff[i], aso, nrA);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (i * NW + wid < B_Q) bufdma16(baseW, sB + (i * NW + wid) * 1024, bvoff[i], bso, nrB);
    if (live) { asoff += 128u * KG; bsoff += 128u * KG; --seg_left; }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (row&7 == lane&7 because every row base is a multiple of 16)
  const int frow = lane & 15;
  int foff[2], foffB[2];                             // A: row = 16i + rho ; B: row = q'*4NI + 4j + r' (rho = 4q' + r')
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int sw = (((kk << 2) + (lane >> 4)) ^ (lane & 7)) << 4;
    foff[kk] = frow * 128 + sw;
    foffB[kk] = (frow >> 2) * (4 * NI * 128) + (frow & 3) * 128 + sw;
  }

  // ---- main loop (2-stage ring): ONE basic block per K step, hand-interleaved so the LDS-DMA issue of
  // the next stage hides under this stage's MFMAs inside the same wave:
  //   [barrier] kk0 fragment reads | { 2 MFMA, 1 LDS-DMA, 1 kk1 fragment read } x n | remaining MFMAs
  // Memory ops keep source order (the compiler cannot prove DMA stores and fragment loads disjoint);
  // sched_group_barrier pins the register-only MFMAs in between.  The last step issues its DMAs
  // through a zero-record descriptor (dropped) so there is no branch in the loop body.
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile rows must split evenly over the DMA lanes");
  constexpr int NF = MI + NI, ND = A_IT + B_IT, NG = NF > ND ? NF : ND;

  // one K step: barrier, then {kk0 fragment reads | (2 MFMA, 1 DMA, 1 kk1 read) x n | remaining MFMAs}
  auto kstep = [&](int cur, int dst, const bf16_t* baseA, const unsigned (&av)[A_IT], unsigned aso, unsigned bso, unsigned nrA, unsigned nrB) {
    char* dA = ring + dst * STAGE;
    char* dB = dA + A_BYTES;
    const char* sA = ring + cur * STAGE + wm * WTM * 128;
    const char* sB = ring + cur * STAGE + A_BYTES + wn * WTN * 128;
    bf16x8 a0[MI], b0[NI], a1[MI], b1[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) a0[i] = *(const bf16x8*)(sA + i * 2048 + foff[0]);
#pragma unroll
    for (int j = 0; j < NI; ++j) b0[j] = *(const bf16x8*)(sB + j * 512 + foffB[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#if defined(AGD_EXPERIMENTS) && defined(EXP_SKIP_DMA)   // timing only: 1 = no A pieces, 2 = no B pieces, 3 = none (results are garbage)
      if (g < A_IT) { if (!(EXP_SKIP_DMA & 1)) bufdma16(baseA, dA + (g * NW + wid) * 1024, av[g < A_IT ? g : 0], aso, nrA); }
      else if (g < ND) { if (!(EXP_SKIP_DMA & 2)) bufdma16(baseW, dB + ((g - A_IT) * NW + wid) * 1024, bvoff[(g >= A_IT && g < ND) ? g - A_IT : 0], bso, nrB); }
#else
      if (g < A_IT) bufdma16(baseA, dA + (g * NW + wid) * 1024, av[g < A_IT ? g : 0], aso, nrA);
      else if (g < ND) bufdma16(baseW, dB + ((g - A_IT) * NW + wid) * 1024, bvoff[(g >= A_IT && g < ND) ? g - A_IT : 0], bso, nrB);
#endif
      if (g < MI) a1[g] = *(const bf16x8*)(sA + g * 2048 + foff[1]);
      else if (g < NF) b1[g - MI] = *(const bf16x8*)(sB + (g - MI) * 512 + foffB[1]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);   // D = W . X^T
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
    // pin the interleave (masks: MFMA 0x8, VMEM_READ 0x20, DS_READ 0x100)
    __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
      if (g < ND) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
      if (g < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x8, 2 * MI * NI - 2 * NG, 0);
  };

  // LayerNorm-fold consumer: one thread per tile row sums the producer's partial sums while the prologue stages are in flight and
  // leaves (mean, rstd) in LDS behind the ring (the epilogue then reads two floats per row instead of chasing `slots` loads)
  float* const lnst = (float*)(smem + KG * (STAGES * STAGE + 4096));
  if (!SPLITK && p.ln_stats && threadIdx.x < BM) {
    const int m = m0 + (int)threadIdx.x;
    float S = 0.f, Q = 0.f;
    if (m < p.M) for (int k = 0; k < p.ln_slots; ++k) { const f32x2 v = *(const f32x2*)(p.ln_stats + ((long long)m * p.ln_slots + k) * 2); S += v[0]; Q += v[1]; }
    const float mu = S * p.ln_invC;
    float var = Q * p.ln_invC - mu * mu; var = var < 0.f ? 0.f : var;
    *(f32x2*)(lnst + threadIdx.x * 2) = f32x2{mu, rsqrtf(var + p.ln_eps)};
  }
  {
    // STAGES-deep ring: stages ks+1 .. ks+STAGES-2 stay in flight across the barrier (counted vmcnt), the stage
    // for step ks+STAGES-1 is issued inside step ks into the slot step ks-1 just released.  A deeper ring is what
    // covers the L2/HBM latency when few workgroups share a CU (small feature maps).
    if (nk > 0 || KG > 1) {
#pragma unroll
      for (int s_ = 0; s_ < STAGES - 1; ++s_) issue(s_, s_ < nk);
    }
    // (K groups: both groups pass the same number of barriers -- group 1 of an odd K runs one step on a dead slot: its loads were issued through
    //  zero-record descriptors, the hardware wrote zeros, the MFMAs add nothing)
    const int nloop = KG == 1 ? nk : (nk_total + KG - 1) / KG;
    for (int ks = 0; ks < nloop; ++ks) {
      wait_vmcnt<(STAGES - 2) * LPS>();
      asm volatile("s_barrier" ::: "memory");
      const int nxt = ks + STAGES - 1;
      const bool more = nxt < nk;
      if (more && seg_left == 0) new_segment();
      const bf16_t* baseA = __builtin_amdgcn_readfirstlane(cursrc) ? base1 : base0;
      const unsigned aso = __builtin_amdgcn_readfirstlane(asoff), bso = __builtin_amdgcn_readfirstlane(bsoff);
      kstep(ks % STAGES, nxt % STAGES, baseA, avoff, aso, bso, more ? nrecA : 0u, more ? nrecB : 0u);
      if (more) { asoff += 128u * KG; bsoff += 128u * KG; --seg_left; }
    }
  }

  // The tail K steps issued their LDS-DMAs through a zero-record descriptor: nothing is fetched, but the hardware still
  // WRITES ZEROS to the LDS destination.  Let those land before the wave retires (the epilogue itself no longer touches LDS).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AGD_EXPERIMENTS
  if (p.dbg & 64) {                                // timing experiment (AGD_IGEMM_CFG=1024): no epilogue (keeps acc live)
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc == 1.2345e-30f) ((float*)p.out)[0] = sacc;
    return;
  }
#endif
  if constexpr (KG == 2) {
    // group 1's accumulators -> LDS (the rings are free: every wave has left the K loop) -> group 0 adds them and finishes the tile; group 1
    // only keeps the epilogue's workgroup barriers company
    __syncthreads();
    f32x4* xch = (f32x4*)smem + (wid * MI * NI) * 64 + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) xch[(i * NI + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) { igemm_epilogue_ghost(p); return; }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] += xch[(i * NI + j) * 64];
  }
  igemm_epilogue<BM, BN, WM, WN, GEGLU, SPLITK>(p, acc, smem, lane, wm, wn, m0, n0, tn, bz, (!SPLITK && p.ln_stats && nk_total > 0) ? lnst : nullptr);
}

// split-K second pass: out = epilogue(sum_s partial[s])   (deterministic slab sum, no atomics)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmP p, int S) {
  const int HWo = p.Hout * p.Wout;
  const long long total = (long long)p.M * (p.N >> 2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / (p.N >> 2)), n = (int)(i % (p.N >> 2)) * 4;
    f32x4 a = *(const f32x4*)(p.splitk_ws + (long long)m * p.N + n);
    for (int s = 1; s < S; ++s) a += *(const f32x4*)(p.splitk_ws + ((long long)s * p.M + m) * p.N + n);
    float v[4] = {a[0], a[1], a[2], a[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (p.bias_mode == 1) v[e] += p.bias[n + e]; else if (p.bias_mode == 2) v[e] += p.bias[m];
      if (p.rowadd) v[e] += p.rowadd[(long long)(m / HWo) * p.rowadd_ld + n + e];
      if (p.residual) v[e] += bf2f(p.residual[(long long)m * p.ldr + n + e]);
      if (p.act == 1) v[e] = silu_f(v[e]); else if (p.act == 2) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e])); else if (p.act == 3) v[e] = gelu_erf_f(v[e]);
    }
    if (p.out_f32) { float* op = (float*)p.out + (long long)m * p.ldo + n; for (int e = 0; e < 4; ++e) op[e] = v[e]; }
    else { bf16_t* op = (bf16_t*)p.out + (long long)m * p.ldo + n; u32x2 pk; pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]);
           if ((p.ldo & 3) == 0) *(u32x2*)op = pk; else for (int e = 0; e < 4; ++e) op[e] = f2bf(v[e]); }
  }
}

// split-K second pass fused with the GroupNorm(+SiLU) that reads the result (IgemmP::gn_y): one workgroup per (image, group) owns the
// [HW][C / groups] block of the output -- sums its slab rows in slice order (deterministic), applies the conv epilogue, rounds to bf16,
// takes the group's mean / variance from the ROUNDED values it holds in registers (what the separate statistics kernel would read back)
// and writes silu?((x - mean) rstd gamma + beta).  Replaces three launches (slab sum, gn_stats, gn_apply) that each re-read the last
// one's output; MAXQ = float4 quads per thread.  The pass is a chain of memory round trips (slabs -> epilogue operands -> statistics ->
// stores) on 256 workgroups, so it runs 1024 threads per workgroup (16 waves per CU in flight instead of 4) and issues EVERY load it
// will need -- slabs in chunks of SU, residual, bias, row add, gamma, beta -- before the first use.
template <int MAXQ, int SU>
__global__ __launch_bounds__(1024) void splitk_reduce_gn_kernel(const IgemmP p, int S) {
  constexpr int NT = 1024;
  __shared__ float red[2][NT / 64];
  const int HWo = p.Hout * p.Wout, cpg = p.N / p.gn_groups, nq = cpg >> 2;
  const int grp = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  const int total = HWo * nq;
  const int n0 = grp * cpg;
  const long long slab = (long long)p.M * p.N;
  // indices of the tail quads are clamped, only their stores are predicated: a branch around each load would make the compiler wait
  // for every one of them in turn
  long long off[MAXQ]; int nn[MAXQ]; bool ok[MAXQ];
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const int qi0 = tid + k * NT;
    ok[k] = qi0 < total;
    const int qi = ok[k] ? qi0 : total - 1;
    const int px = qi / nq;
    nn[k] = n0 + 4 * (qi - px * nq);
    off[k] = ((long long)img * HWo + px) * p.N + nn[k];
  }
  f32x4 a[MAXQ];
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) a[k] = *(const f32x4*)(p.splitk_ws + off[k]);
  // epilogue operands: requested now, used after the slab sums (absent ones read a valid dummy address and are not added)
  const bf16_t* rbase = p.residual ? p.residual : (const bf16_t*)p.splitk_ws;
  const int ldr = p.residual ? p.ldr : p.N;
  const float* bbase = p.bias_mode == 1 ? p.bias : p.gn_gamma;
  const float* abase = p.rowadd ? p.rowadd + (long long)img * p.rowadd_ld : p.gn_gamma;
  u32x2 rr[MAXQ]; f32x4 bb[MAXQ], ra[MAXQ], ga[MAXQ], be[MAXQ];
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const long long m = (off[k] - nn[k]) / p.N;
    rr[k] = *(const u32x2*)(rbase + m * ldr + nn[k]);
    bb[k] = *(const f32x4*)(bbase + nn[k]);
    ra[k] = *(const f32x4*)(abase + nn[k]);
    ga[k] = *(const f32x4*)(p.gn_gamma + nn[k]);
    be[k] = *(const f32x4*)(p.gn_beta + nn[k]);
  }
  for (int s = 1; s < S; s += SU) {
    f32x4 t[SU][MAXQ];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int sc = s + u < S ? s + u : S - 1;
#pragma unroll
      for (int k = 0; k < MAXQ; ++k) t[u][k] = *(const f32x4*)(p.splitk_ws + sc * slab + off[k]);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u)
      if (s + u < S) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) a[k] += t[u][k];
      }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const int n = nn[k];
    const long long m = (off[k] - n) / p.N;
    if (p.bias_mode == 1) a[k] += bb[k]; else if (p.bias_mode == 2) a[k] += p.bias[m];
    if (p.rowadd) a[k] += ra[k];
    if (p.residual) {
      const u32x2 r = rr[k];
      a[k] += f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xFFFF0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xFFFF0000u)};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = a[k][e];
      if (p.act == 1) x = silu_f(x);
      x = bf2f(f2bf(x));                                  // the stored activation: statistics and normalisation see the rounded value
      a[k][e] = x;
      if (ok[k]) { s1 += x; s2 += x * x; }
    }
    if (p.gn_keep_out && ok[k]) {
      u32x2 pk; pk[0] = pack_bf2(a[k][0], a[k][1]); pk[1] = pack_bf2(a[k][2], a[k][3]);
      *(u32x2*)((bf16_t*)p.out + m * p.ldo + n) = pk;
    }
  }
  for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
  __syncthreads();
  const double cnt = (double)HWo * cpg;
  double S1 = 0.0, S2 = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { S1 += (double)red[0][w]; S2 += (double)red[1][w]; }      // fixed order
  const double mean = S1 / cnt;
  double var = S2 / cnt - mean * mean; if (var < 0) var = 0;
  const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sc = rstd * ga[k][e];                   // same affine form as gn_apply: x * scale + shift
      const float r = fmaf(a[k][e], sc, be[k][e] - mu * sc);
      o[e] = p.gn_silu ? silu_f(r) : r;
    }
    if (ok[k]) { u32x2 pk; pk[0] = pack_bf2(o[0], o[1]); pk[1] = pack_bf2(o[2], o[3]); *(u32x2*)(p.gn_y + off[k]) = pk; }
  }
}
// the shapes the fused slab-sum + GroupNorm pass takes (else the caller's separate GroupNorm launches run)
static bool reduce_gn_ok(const IgemmP& p) {
  if (!p.gn_y || !p.gn_gamma || !p.gn_beta || p.gn_groups < 1 || p.out_f32 || p.geglu || p.batch > 1) return false;
  const int HWo = p.Hout * p.Wout;
  if (p.N % p.gn_groups || (p.N / p.gn_groups) % 4 || p.M % HWo || p.ldo != p.N || (p.residual && p.ldr % 4)) return false;
  return (long long)HWo * (p.N / p.gn_groups / 4) <= 1024 * 3 && p.act <= 1;
}

// the slab pass of a split-K launch: out = epilogue(sum of the S slabs), fused with the GroupNorm that reads it where the caller asked for it
static int launch_splitk_reduce(const IgemmP& p, int splits, hipStream_t st) {
  if (reduce_gn_ok(p)) {
    const long long quads = (long long)p.Hout * p.Wout * (p.N / p.gn_groups / 4);
    const dim3 g(p.gn_groups, p.M / (p.Hout * p.Wout));
    // slab loads per round: all of a 2-way split at once; the 8 x 8 maps' 8 - 10 slabs (one quad per thread) all at once
    if (quads <= 1024) hipLaunchKernelGGL((splitk_reduce_gn_kernel<1, 9>), g, dim3(1024), 0, st, p, splits);
    else if (quads <= 2048) hipLaunchKernelGGL((splitk_reduce_gn_kernel<2, 2>), g, dim3(1024), 0, st, p, splits);
    else hipLaunchKernelGGL((splitk_reduce_gn_kernel<3, 1>), g, dim3(1024), 0, st, p, splits);
    HIP_CHECK_RET(hipGetLastError());
    if (p.gn_fused) *p.gn_fused = 1;
    return 0;
  }
  const long long total = (long long)p.M * (p.N >> 2);
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, p, splits);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// XCD-aware tile blocks (IgemmP::xb_m / xb_n, tile_of in common.h): the eight XCDs each get an a x b block of the tile grid instead of q = tiles / 8 consecutive tiles of
// the A-major (or W-major) walk, (a, b) minimising the XCD's working set a x (A panel) + b x (W panel).  tools/ubench/l2_stride.hip: with the XCD's working set inside its
// 4 MB L2 a CU's LDS-DMA stream runs at 113 GB/s beside the tile's MFMAs, beyond it at 79 GB/s -- and a row of tiles is the largest working set q tiles can have
// (M = 2048, K = N = 1280 on 64 x 160 tiles: 4 x 8 tiles = 0.66 + 3.3 MB of operands per XCD, 8 x 4 tiles = 1.3 + 1.6 MB).
static void pick_xcd_block(IgemmP& p, int BM, int BN) {
  p.xb_m = p.xb_n = 0;
  if (!p.xcd_block || p.batch > 1 || p.w_per_image) return;
  const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
  const long long T = (long long)tm * tn;
  if (T % 8 || T < 16 || T > (1 << 20)) return;
  const int q = (int)(T / 8);
  const double ca = (double)BM * (p.C0 + p.C1), cw = (double)BN * p.K;      // elements of one A panel (unique pixels) / one W panel
  double best = 1e300; int ba = 0, bb = 0;
  for (int a = 1; a <= q; ++a) {
    if (q % a) continue;
    const int b = q / a;
    if (tm % a || tn % b) continue;
    const double cost = a * ca + b * cw;
    if (cost < best) { best = cost; ba = a; bb = b; }
  }
  if (!ba) return;
  // the walk's own shape: q consecutive ids of an A-major walk cover ceil(q / tn) (+1 when they straddle) rows of tiles x min(q, tn) columns; W-major alike
  double def;
  if (!p.wmajor) { const int rows = q >= tn ? (q + tn - 1) / tn + ((q % tn) ? 1 : 0) : ((tn % q) ? 2 : 1); def = rows * ca + (q >= tn ? tn : q) * cw; }
  else { const int cols = q >= tm ? (q + tm - 1) / tm + ((q % tm) ? 1 : 0) : ((tm % q) ? 2 : 1); def = cols * cw + (q >= tm ? tm : q) * ca; }
  if (best < 0.9 * def) { p.xb_m = ba; p.xb_n = bb; if (p.warm == 1) p.warm = 2; }      // (the W-major warm-up streams the walk's per-XCD row range: with blocks, the whole matrix once)
}

template <int BM, int BN, int WM, int WN, int KS, int STAGES, int GEGLU, int SPLITK, int KG = 1>
static int launch_one(const IgemmP& p_in, int splits, hipStream_t st) {
  IgemmP p = p_in; pick_xcd_block(p, BM, BN);
  constexpr int NT = WM * WN * 64 * KG;
  constexpr int stage = (BM + BN) * 128;
  constexpr int lds = KG * (STAGES * stage + 4096) + BM * 8;    // per K group: the ring + one scratch KiB per wave (cold-weight warm-up pieces); + (mean, rstd) of the tile's rows
  static_assert(lds <= 160 * 1024, "LDS");
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  dim3 grid(tiles, p.batch > 0 ? p.batch : 1, splits);
  auto kfn = igemm_kernel<BM, BN, WM, WN, KS, STAGES, GEGLU, SPLITK, KG>;
  // the dynamic-LDS attribute is per device: latch it per (instantiation, device)
  static std::atomic<bool> attr[AGD_MAX_DEVICES] = {};
  int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
  if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("igemm: device ordinal %d out of range", dev); return -1; }
  if (!attr[dev]) { HIP_CHECK_RET(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr[dev] = true; }
  hipLaunchKernelGGL(kfn, grid, dim3(NT), lds, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// row-halo 3x3 kernel (igemm_halo.h): stride 1, pad 1, optional nearest-2x upsample, tile rows = whole output rows (or 128-pixel segments)
static bool halo_ok(const IgemmP& p) {
  return p.halo && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.Hin * p.up == p.Hout && p.Win * p.up == p.Wout && p.batch <= 1 &&
         !p.geglu && p.Wout >= 16 && (p.Wout <= 128 ? 128 % p.Wout == 0 : p.Wout % 128 == 0) &&      // 8x8 maps (8-way split-K) measured 4 % slower
         (long long)p.Hin * p.Win * (p.C0 > p.C1 ? p.C0 : p.C1) * (p.M / (p.Hin * p.Win) + 1) < (1LL << 30);   // 32-bit byte offsets per source
}
template <int BN, int SPLITK, int BST>
static int launch_halo(const IgemmP& p_in, int splits, hipStream_t st) {
  IgemmP p = p_in; pick_xcd_block(p, 128, BN);
  const int Wt = p.Wout < 128 ? p.Wout : 128;
  const int hr = (128 / Wt) * (Wt + 2), HRP = (hr + 7) & ~7;
  const int lds = 2 * HRP * 128 + BST * BN * 128 + 4096;   // two A images of the tile's halo rows + the weight ring + the dead-piece sink
  const int tiles = ((p.M + 127) / 128) * ((p.N + BN - 1) / BN);
  auto kfn = igemm_halo_kernel<BN, SPLITK, BST>;
  static std::atomic<bool> attr[AGD_MAX_DEVICES] = {};
  int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
  if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("igemm: device ordinal %d out of range", dev); return -1; }
  if (!attr[dev]) { HIP_CHECK_RET(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 160 * 128 + BST * BN * 128 + 4096)); attr[dev] = true; }
  hipLaunchKernelGGL(kfn, dim3(tiles, 1, splits), dim3(256), lds, st, p, Wt, HRP);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// producer / consumer row-halo kernel (igemm_pch.h): 128 x 160 tiles, one workgroup per CU (IgemmP::pc bit 4)
static bool pch_ok(const IgemmP& p, int splits) {
  if (!(p.pc & 16) || !halo_ok(p) || p.up != 1 || p.w_per_image) return false;
  const long long tiles = (long long)((p.M + 127) / 128) * ((p.N + 159) / 160);
  return tiles * splits <= 256;
}
template <int SPLITK>
static int launch_pch(const IgemmP& p_in, int splits, hipStream_t st) {
  IgemmP p = p_in; pick_xcd_block(p, 128, 160);
  const int Wt = p.Wout < 128 ? p.Wout : 128;
  const int hr = (128 / Wt) * (Wt + 2), HRP = (hr + 7) & ~7;
  const int lds = 3 * HRP * 128 + 5 * 160 * 128 + 4 * 1024;      // three A images (Wout >= 16: at most 144 halo rows) + five weight stages + the dead-piece sink: <= 158 KB
  const int tiles = ((p.M + 127) / 128) * ((p.N + 159) / 160);
  auto kfn = igemm_pch_kernel<160, SPLITK>;
  static std::atomic<bool> attr[AGD_MAX_DEVICES] = {};
  int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
  if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("igemm_pch: device ordinal %d out of range", dev); return -1; }
  if (!attr[dev]) { HIP_CHECK_RET(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 144 * 128 + 5 * 160 * 128 + 4 * 1024)); attr[dev] = true; }
  hipLaunchKernelGGL(kfn, dim3(tiles, 1, splits), dim3(512), lds, st, p, Wt, HRP);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <int BM, int BN, int WM, int WN, int STAGES = 2, int KG = 1, int KGS = 2>      // KGS: ring slots per K group when KG = 2
static int launch_cfg(const IgemmP& p, int splits, hipStream_t st) {
  if (p.cfg_out) { p.cfg_out[0] = BM; p.cfg_out[1] = BN; p.cfg_out[2] = splits; return 0; }   // igemm_query: report the dispatch decision only
  if ((p.rowstat_out || p.ln_stats) && splits > 1) { agd_set_error("igemm: LayerNorm fold on a split-K launch"); return -1; }
  if (p.colstat_out && (splits > 1 || p.colstat_rows < 1 || p.colstat_rows % BM)) { agd_set_error("igemm: column statistics need an unsplit launch whose M tiles stay inside one image"); return -1; }
  if (p.rowstat_out && p.rowstat_slots != (p.N + BN - 1) / BN) { agd_set_error("igemm: rowstat_slots %d != N tiles %d", p.rowstat_slots, (p.N + BN - 1) / BN); return -1; }
#ifdef AGD_EXPERIMENTS
  static const bool logit = getenv("AGD_IGEMM_LOG") != nullptr;      // tools/layer_report.py joins this with a kernel trace
  if (logit) fprintf(stderr, "IGEMM M=%d N=%d K=%d ks=%d stride=%d up=%d geglu=%d res=%d tile=%dx%d splits=%d batch=%d\n", p.M, p.N, p.K, p.ksize,
                     p.stride, p.up, p.geglu, p.residual ? 1 : 0, BM, BN, splits, p.batch > 0 ? p.batch : 1);
#endif
  constexpr bool HALO_TILE = BM == 128 && WM == 2 && WN == 2 && (BN == 128 || BN == 160);
  if (p.sc0 && !HALO_TILE) { agd_set_error("igemm: shortcut fusion on a launch that is not a row-halo launch (tile %dx%d, %d K slices)", BM, BN, splits); return -1; }
  if (splits > 1) {
    int rc;
    if constexpr (HALO_TILE) { rc = (BN == 160 && pch_ok(p, splits)) ? launch_pch<1>(p, splits, st) : halo_ok(p) ? launch_halo<BN, 1, STAGES == 4 ? 4 : 2>(p, splits, st) : (p.ksize == 3) ? launch_one<BM, BN, WM, WN, 3, STAGES, 0, 1>(p, splits, st) : launch_one<BM, BN, WM, WN, 1, STAGES, 0, 1>(p, splits, st); }
    else rc = (p.ksize == 3) ? launch_one<BM, BN, WM, WN, 3, STAGES, 0, 1>(p, splits, st) : launch_one<BM, BN, WM, WN, 1, STAGES, 0, 1>(p, splits, st);
    if (rc) return rc;
    return launch_splitk_reduce(p, splits, st);
  }
  if (p.geglu) {
    if constexpr (BN / WN == 64) { if (p.ksize == 1) return launch_one<BM, BN, WM, WN, 1, STAGES, 1, 0>(p, 1, st); }
    agd_set_error("igemm: geglu only on 1x1 with the 128-wide tile"); return -1;
  }
  if constexpr (HALO_TILE) {
    if (BN == 160 && pch_ok(p, 1)) return launch_pch<0>(p, 1, st);
    if (halo_ok(p)) return launch_halo<BN, 0, STAGES == 4 ? 4 : 2>(p, 1, st);
  }
  if (p.ksize == 3) return launch_one<BM, BN, WM, WN, 3, STAGES, 0, 0>(p, 1, st);
  if (p.ksize == 2) {
    if constexpr (BM == 128 && STAGES == 2 && KG == 1) return launch_one<BM, BN, WM, WN, 2, 2, 0, 0>(p, 1, st);
    if constexpr (BM == 64 && BN == 160 && STAGES == 4 && KG == 1) return launch_one<64, 160, WM, WN, 2, 4, 0, 0>(p, 1, st);
    agd_set_error("igemm: 2x2 phase convs run on the 128-row two-stage tiles or the 64 x 160 deep-ring tile"); return -1;
  }
  if constexpr (KG == 2) {
    // two K groups of waves per workgroup: plain single-source 1x1 launches (the kernel's K walk has one segment)
    if (p.C1 == 0 && p.stride == 1 && p.up == 1 && p.pad == 0 && p.Hin == p.Hout && p.Win == p.Wout && (p.batch <= 1) && !p.w_per_image)
      return launch_one<BM, BN, WM, WN, 1, KGS, 0, 0, 2>(p, 1, st);
  }
  return launch_one<BM, BN, WM, WN, 1, STAGES, 0, 0>(p, 1, st);
}

// split-K partial slabs: the caller's (per-ctx) workspace when it supplies one, else one per device for the ctx-less
// single-op entry points.  Reuse is stream-ordered: every launch that writes the slabs is followed, on the same stream,
// by the reduce that reads them.
static SplitKWs g_splitk_dev[AGD_MAX_DEVICES];
static int ensure_splitk(IgemmP& p, int S) {
  SplitKWs* ws = p.ws;
  if (!ws) {
    int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("igemm: device ordinal %d out of range", dev); return -1; }
    ws = &g_splitk_dev[dev];
  }
  const size_t need = (size_t)S * p.M * p.N * 4;
  if (need > ws->cap) {
    if (ws->p) { (void)hipDeviceSynchronize(); (void)hipFree(ws->p); }
    const size_t cap = need > ((size_t)64 << 20) ? need : ((size_t)64 << 20);
    if (hipMalloc((void**)&ws->p, cap) != hipSuccess) { ws->p = nullptr; ws->cap = 0; agd_set_error("split-K workspace alloc failed"); return -1; }
    ws->cap = cap;
  }
  p.splitk_ws = ws->p;
  return 0;
}

// ---- producer / consumer kernel (igemm_pc.h): loader waves + consumer waves, one workgroup per CU
static bool pc_ok(const IgemmP& p) {
  const bool lin = p.ksize == 1 && p.stride == 1 && p.up == 1 && p.pad == 0 && p.Hin == p.Hout && p.Win == p.Wout;
  const bool c33 = p.ksize == 3 && p.stride == 1 && p.up == 1 && p.pad == 1 && p.Hin == p.Hout && p.Win == p.Wout && !p.sc0;
  if (!(lin || c33) || p.batch > 1 || p.geglu || p.ups4) return false;
  const long long amax = (long long)p.M * (p.C0 > p.C1 ? p.C0 : p.C1) * 2;
  return p.M < (1 << 24) && amax < (1LL << 31) && (long long)p.N * p.K * 2 < (1LL << 31);
}
template <int BM, int BN, int NLW, int STAGES>
static int launch_pc(const IgemmP& p, int splits, hipStream_t st) {
  using G = PcGeom<BM, BN, NLW, STAGES>;
  if (p.cfg_out) { p.cfg_out[0] = BM; p.cfg_out[1] = BN; p.cfg_out[2] = splits; return 0; }
  if ((p.rowstat_out || p.ln_stats) && splits > 1) { agd_set_error("igemm_pc: LayerNorm fold on a split-K launch"); return -1; }
  if (p.colstat_out && (splits > 1 || p.colstat_rows < 1 || p.colstat_rows % BM)) { agd_set_error("igemm_pc: column statistics need an unsplit launch whose M tiles stay inside one image"); return -1; }
  if (p.rowstat_out && p.rowstat_slots != (p.N + BN - 1) / BN) { agd_set_error("igemm_pc: rowstat_slots %d != N tiles %d", p.rowstat_slots, (p.N + BN - 1) / BN); return -1; }
#ifdef AGD_EXPERIMENTS
  static const bool logit = getenv("AGD_IGEMM_LOG") != nullptr;
  if (logit) fprintf(stderr, "IGEMM M=%d N=%d K=%d ks=%d stride=%d up=%d geglu=%d res=%d tile=%dx%d splits=%d batch=1\n", p.M, p.N, p.K, p.ksize,
                     p.stride, p.up, p.geglu, p.residual ? 1 : 0, BM, BN, splits);
#endif
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const void* kfn = p.ksize == 3 ? (splits > 1 ? (const void*)igemm_pc_kernel<BM, BN, NLW, STAGES, 3, 0, 1> : (const void*)igemm_pc_kernel<BM, BN, NLW, STAGES, 3, 0, 0>)
                                 : (splits > 1 ? (const void*)igemm_pc_kernel<BM, BN, NLW, STAGES, 1, 0, 1> : (const void*)igemm_pc_kernel<BM, BN, NLW, STAGES, 1, 0, 0>);
  static std::atomic<bool> attr[AGD_MAX_DEVICES][4] = {};
  int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
  if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("igemm_pc: device ordinal %d out of range", dev); return -1; }
  const int slot = (p.ksize == 3 ? 2 : 0) + (splits > 1 ? 1 : 0);
  if (!attr[dev][slot]) { HIP_CHECK_RET(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS)); attr[dev][slot] = true; }
  IgemmP pp = p; pick_xcd_block(pp, BM, BN);
  void* args[] = {&pp};
  HIP_CHECK_RET(hipLaunchKernel(kfn, dim3((unsigned)tiles, 1, (unsigned)splits), dim3(G::THREADS), args, G::LDS, st));
  if (splits > 1) return launch_splitk_reduce(pp, splits, st);
  return 0;
}

// ---- 8-wave / 8-phase kernel (igemm8p.h): 256 x 256 tiles (2 x 4 waves) and 256 x 160 tiles (4 x 2 waves)
template <int WM, int WN, int MI, int NI0, int NI1>
static int launch_8p(const IgemmP& p, hipStream_t st) {
  using G = P8Geom<WM, WN, MI, NI0, NI1>;
  if (p.cfg_out) { p.cfg_out[0] = G::BM; p.cfg_out[1] = G::BN; p.cfg_out[2] = 1; return 0; }
  if (p.colstat_out && (p.colstat_rows < 1 || p.colstat_rows % G::BM)) { agd_set_error("igemm8p: column statistics need M tiles inside one image"); return -1; }
  if (p.rowstat_out && p.rowstat_slots != (p.N + G::BN - 1) / G::BN) { agd_set_error("igemm8p: rowstat_slots %d != N tiles %d", p.rowstat_slots, (p.N + G::BN - 1) / G::BN); return -1; }
#ifdef AGD_EXPERIMENTS
  static const bool logit = getenv("AGD_IGEMM_LOG") != nullptr;
  if (logit) fprintf(stderr, "IGEMM M=%d N=%d K=%d ks=%d stride=%d up=%d geglu=%d res=%d tile=%dx%d splits=1 batch=1\n", p.M, p.N, p.K, p.ksize, p.stride, p.up,
                     p.geglu, p.residual ? 1 : 0, G::BM, G::BN);
#endif
  const int tiles = ((p.M + G::BM - 1) / G::BM) * ((p.N + G::BN - 1) / G::BN);
  const void* kfn = nullptr;
  if (p.geglu) {
    if constexpr (G::NI == 4) kfn = (const void*)igemm8p_kernel<WM, WN, MI, NI0, NI1, 1, 1>;
    if (!kfn || p.ksize != 1) { agd_set_error("igemm8p: geglu only on 1x1 with the 256-wide tile"); return -1; }
  } else kfn = p.ksize == 3 ? (const void*)igemm8p_kernel<WM, WN, MI, NI0, NI1, 3, 0> : p.ksize == 2 ? (const void*)igemm8p_kernel<WM, WN, MI, NI0, NI1, 2, 0> : (const void*)igemm8p_kernel<WM, WN, MI, NI0, NI1, 1, 0>;
  static std::atomic<bool> attr[AGD_MAX_DEVICES][4] = {};
  int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
  if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("igemm8p: device ordinal %d out of range", dev); return -1; }
  const int slot = p.geglu ? 2 : p.ksize == 3 ? 1 : p.ksize == 2 ? 3 : 0;
  if (!attr[dev][slot]) { HIP_CHECK_RET(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS)); attr[dev][slot] = true; }
  IgemmP pp = p; pick_xcd_block(pp, G::BM, G::BN);
  void* args[] = {&pp};
  HIP_CHECK_RET(hipLaunchKernel(kfn, dim3(tiles), dim3(512), args, G::LDS, st));
  return 0;
}
// 0: not for this launch; 1: 256 x 256 tiles; 2: 256 x 160 tiles.  One workgroup per CU, no split-K: the launch must bring enough
// tiles, and the share of MFMA slots doing useful work (tile padding in M and N x the last, partial wave of tiles) decides.
static int pick_8p(const IgemmP& p) {
  const bool lin = p.ksize == 1 && p.stride == 1 && p.up == 1 && p.pad == 0 && p.Hin == p.Hout && p.Win == p.Wout;
  if (!(p.ksize == 3 || lin) || p.batch > 1 || p.rowstat_out) return 0;      // (the 8-phase epilogue carries no LayerNorm row-statistics producer)
  if (p.M >= (1 << 24) || (long long)p.N * p.K * 2 >= (1LL << 31)) return 0;
  const long long amax = (long long)p.Hin * p.Win * (p.C0 > p.C1 ? p.C0 : p.C1) * 2;       // bytes of one image of the wider source
  if (amax * ((long long)p.M / (p.Hout * p.Wout) + 1) >= (1LL << 31)) return 0;                // 32-bit byte offsets per source
  // the 8-phase kernels carry only the vector epilogue: every lane's channel run (16 / 20 wide) must be whole and 16- / 8-byte aligned
  const int nout = p.geglu ? p.N / 2 : p.N;
  auto vec_ok = [&](int run, int sv) { return p.N % run == 0 && p.ldo % (p.out_f32 ? 4 : sv) == 0 && (!p.residual || p.ldr % sv == 0) && nout % sv == 0; };
  const bool okA = vec_ok(16, 8), okB = !p.geglu && vec_ok(20, 4);
  if (p.p8 == 2) return okA ? 1 : 0;
  if (p.p8 == 3) return okB ? 2 : 0;
  auto eff = [&](int bn) {
    const long long t = (long long)((p.M + 255) / 256) * ((p.N + bn - 1) / bn), waves = (t + 255) / 256;
    return (double)p.M * p.N / ((double)waves * 256.0 * 256.0 * bn);
  };
  const double eA = okA ? eff(256) : 0.0, eB = okB ? eff(160) : 0.0;
  if (p.p8 == 4) return (okA || okB) ? (eA >= eB ? 1 : 2) : 0;       // tests: every legal launch
  // Measured against the 4-wave kernels on the SD-1.5 / VAE shapes (tools/kb_8p.py, hot operands): the 256-wide tile wins on wide
  // 1x1 launches (L0 qkv 46.9 -> 37.0 us, L0 GEGLU 92 -> 81, L1 qkv 29.6 -> 26.5, L1 GEGLU 74 -> 70) and on 3x3 convs whose N is a
  // multiple of 256 (VAE 128 px 512 -> 512: 275 -> 248, VAE up-conv 256: 1090 -> 997); the 160-wide tile (12- and 8-MFMA phases) only
  // ties the 128 x 160 row-halo kernel on the N = 320 convs (62.4 vs 63.3, 155 vs 150) and wins on the upsampling conv, which has no
  // whole-row halo image at 256 rows (216 -> 194).  Launches with fewer than ~200 tiles lose everywhere (one workgroup per CU, no split-K).
  if ((p.K >> 6) < 4) return 0;
  if (eA >= 0.78 && (p.ksize == 3 || p.N >= 768)) return 1;
  if (eB >= 0.9 && p.ksize == 3 && p.up == 2) return 2;
  return 0;
}

#ifdef AGD_EXPERIMENTS
int g_igemm_cfg = 0;   // experiment knob (tools/ only; production builds have no run-time dispatch knobs)
extern "C" __attribute__((visibility("default"))) void agd_set_igemm_cfg(int v) { g_igemm_cfg = v; }
#ifdef AGD_STAMPS
// in-kernel time stamps (tools/kb_*_trace.py): pick the workgroup, read the marks back
extern "C" __attribute__((visibility("default"))) int agd_smap_ts(int wg, unsigned long long* out) {
  if (out) return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_smap_ts), 1024 * 8) == hipSuccess ? 0 : -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_smap_ts_wg), &wg, 4) == hipSuccess ? 0 : -1;
}
#endif
#define KNOB(n) ((g_igemm_cfg & 15) == (n))
#else
#define KNOB(n) false
#endif

static bool nosplit_early(const IgemmP& p) { return p.rowstat_out || p.ln_stats || p.colstat_out; }
int launch_igemm(const IgemmP& p_in, hipStream_t st) {
  IgemmP p = p_in;
#ifdef AGD_EXPERIMENTS
  static bool env_read = false;
  if (!env_read) { env_read = true; const char* e = getenv("AGD_IGEMM_CFG"); if (e) g_igemm_cfg = atoi(e); }
  p.dbg = g_igemm_cfg >> 4;
  { static int stg = -1; if (stg < 0) { const char* e = getenv("AGD_IGEMM_STAGGER"); stg = e ? atoi(e) : 0; } p.stagger = stg; }
#endif
  if (p.K & 63 || p.C0 & 63 || p.C1 & 63) { agd_set_error("igemm: K/C0/C1 must be multiples of 64 (K=%d C0=%d C1=%d)", p.K, p.C0, p.C1); return -1; }
  if (p.K != p.ksize * p.ksize * (p.C0 + p.C1) + p.sc_C0 + p.sc_C1) { agd_set_error("igemm: K=%d != ks^2*(C0+C1)+shortcut=%d", p.K, p.ksize * p.ksize * (p.C0 + p.C1) + p.sc_C0 + p.sc_C1); return -1; }
  if (p.sc0) {          // conv_shortcut folded into a 3x3 launch: the unsplit row-halo kernel only (igemm_can_fuse_shortcut tells the caller beforehand)
    const bool smap8 = p.smap && p.stride == 1 && p.pad == 1 && p.up == 1 && p.Hin == 8 && p.Win == 8 && p.Hout == 8 && p.Wout == 8;      // the 8 x 8 whole-images kernel walks the chunks too
    if (p.ksize != 3 || (p.sc_C0 & 63) || (p.sc_C1 & 63) || p.sc_C0 < 64 || (p.sc_C1 && !p.sc1) || !(halo_ok(p) || smap8) || p.w_per_image || p.geglu) { agd_set_error("igemm: shortcut fusion needs a row-halo or 8 x 8 whole-images 3x3 launch and 64-channel multiples"); return -1; }
    p.p8 = 0; if (!smap8) p.smap = 0;
  }
  if (p.ksize != 1 && p.ksize != 3 && !(p.ksize == 2 && p.ups4)) { agd_set_error("igemm: ksize %d", p.ksize); return -1; }
  if (p.ups4) {          // phase-decomposed upsampling conv: the general 4-wave kernel on 128-row tiles that stay inside one phase
    if (p.ksize != 2 || p.stride != 1 || p.up != 1 || p.Hin != p.Hout || p.Win != p.Wout || p.N != 4 * p.ups4 || p.C1 || p.geglu || p.residual || p.rowadd || p.out_f32 || p.w_per_image ||
        p.sc0 || (p.batch > 1) || p.ldo != p.ups4 || (p.ups4 % 160 && p.ups4 % 128) || p.rowstat_out || p.ln_stats) { agd_set_error("igemm: unsupported phase-conv launch"); return -1; }
    const int p8 = p.p8;
    p.p8 = 0; p.smap = 0; p.halo = 0; p.warm = 0;
    // the 8-phase kernel where its 256-row tiles fill the chip and stay inside one phase (160- or 256-wide), else the 4-wave general kernel
    if (p8 && p.M % 256 == 0 && p.M < (1 << 24) && (long long)p.N * p.K * 2 < (1LL << 31) && (long long)p.Hin * p.Win * p.C0 * 2 * (p.M / (p.Hout * p.Wout) + 1) < (1LL << 31)) {
      if (p.ups4 % 256 == 0 && (long long)(p.M / 256) * (p.N / 256) >= 256) return launch_8p<2, 4, 8, 2, 2>(p, st);
      if (p.ups4 % 160 == 0 && (long long)(p.M / 256) * (p.N / 160) >= 256) return launch_8p<4, 2, 4, 3, 2>(p, st);
    }
    // few source rows (the 8 x 8 -> 16 x 16 upsampler: M = 512): 64 x 160 tiles on the deep ring give one workgroup per CU where the 128-row tiles give half a wave
    if (p.ups4 % 160 == 0 && p.M % 64 == 0 && (long long)(p.M / 128) * (p.N / 160) < 192 && (long long)(p.M / 64) * (p.N / 160) <= 512) return launch_cfg<64, 160, 2, 2, 4>(p, 1, st);
    return (p.ups4 % 160) == 0 ? launch_cfg<128, 160, 2, 2>(p, 1, st) : launch_cfg<128, 128, 2, 2>(p, 1, st);
  }
  if (p.M < 1 || p.N < 1) { agd_set_error("igemm: empty problem M=%d N=%d", p.M, p.N); return -1; }
  if (p.rowadd && p.M >= (1 << 24)) { agd_set_error("igemm: rowadd needs M < 2^24"); return -1; }
  if (p.geglu && (p.N % 128)) { agd_set_error("igemm: geglu needs N %% 128 == 0"); return -1; }
  if ((p.rowstat_out || p.colstat_out) && (p.geglu || p.out_f32 || p.batch > 1)) { agd_set_error("igemm: row / column statistics only for plain bf16 launches"); return -1; }
  if (p.ln_stats && (!p.ln_cs || p.ln_slots < 1 || p.batch > 1)) { agd_set_error("igemm: LayerNorm fold needs colsum + slots"); return -1; }
  const int batch = p.batch > 0 ? p.batch : 1;
  if (p.w_per_image) {           // image i's rows multiply with W + i * sW: plain 1x1 launches whose M tiles stay inside one image, general kernel only
    const int hw = p.Hout * p.Wout;
    if (p.ksize != 1 || batch != 1 || p.geglu || hw % 64 || p.M % hw || p.sW < (long long)p.N * p.K) { agd_set_error("igemm: per-image weights need a plain 1x1 launch with Hout*Wout %% 64 == 0"); return -1; }
    p.p8 = 0; p.warm = 0;
  }
  {  // tile walk order: W-major when the weight matrix is the larger operand (bytes fetched once per XCD either way)
    const double a_bytes = 2.0 * p.M * (double)(p.C0 + p.C1);   // ~ the input pixels (each fetched once from the fabric per XCD)
    const double w_bytes = 2.0 * p.N * (double)p.K;
    // measured (tools/kb_lin.py, AGD_IGEMM_WMAJOR): M=2048 K=1280 GEGLU 79.6 -> 66.8 us, qkv 25.4 -> 24.6; 3x3 convs get SLOWER W-major
    // (72.0 -> 74.6: their tap re-reads of the A rows stop hitting in L2), so 1x1 only
    p.wmajor = (batch == 1 && p.ksize == 1 && w_bytes > 1.5 * a_bytes) ? 1 : 0;
    if (p.warm) {   // 1: W-major, per-XCD slices; 2: A-major with a matrix worth streaming once (p.warm as given: 1 = W-major only, 3 = both)
      const int req = p.warm;
      p.warm = 0;
      if (w_bytes < 2147483648.0 && batch == 1) {
        // measured in situ (tools/ab_option.py weight_warm): W-major only 571.4; + A-major >= 3 MB, M <= 8192: 565.6; M >= 1024 only: ~563;
        // M <= 32768: 560.6; >= 1 MB: -2.7 more; 0.25 .. 1 MB thresholds equal, 2 MB worse; the 8x8 maps (M = 512, split-K 8) lose
        if (p.wmajor && w_bytes >= (double)(1 << 20)) p.warm = 1;
        else if (!p.wmajor && req == 3 && w_bytes >= 1e6 && p.M >= 1024 && p.M <= 32768) p.warm = 2;
      }
    }
#ifdef AGD_EXPERIMENTS
    { static int f = -2; if (f == -2) { const char* e = getenv("AGD_IGEMM_WMAJOR"); f = e ? atoi(e) : -1; } if (f >= 0) p.wmajor = f && batch == 1; }
#endif
  }
  const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
  const int nk = p.K >> 6;
  if (p.w_per_image) {
    // per-image matrices on the small maps (the output GEMM of the pre-multiplied attn2, xattn_pre.hip: M = 2048 / 512, N = 1280, K = 640): 64 x 160 tiles on
    // the deep ring where they make at most one wave of workgroups (256 at 16 x 16); images of fewer than 128 rows (8 x 8 maps) need 64-row tiles in any case
    const int hw = p.Hout * p.Wout;
    if ((p.N % 160) == 0 && (long long)(p.M / 64) * (p.N / 160) <= 256) {
      if ((p.pc & 1) && pc_ok(p)) return launch_pc<64, 160, 4, 5>(p, 1, st);
      return launch_cfg<64, 160, 2, 2, 4>(p, 1, st);
    }
    if (hw % 128) return launch_cfg<64, 64, 2, 2, 4>(p, 1, st);
  }
  // LayerNorm-fold producers / consumers and GroupNorm-statistics producers finish in the tile's own epilogue: they never take a split-K
  // configuration (a transformer wider than SD's -- K >= 4096 at M <= 512 -- falls through to an unsplit tile instead of failing the forward)
  const bool nosplit = p.rowstat_out || p.ln_stats || p.colstat_out;
#ifdef AGD_EXPERIMENTS
  {  // exploration only: AGD_IGEMM_FORCE="<bn>:<splits>:<stages>" forces one configuration for every non-GEGLU launch
    static int f_bn = -1, f_s = 1, f_st = 2;
    if (f_bn < 0) { const char* e = getenv("AGD_IGEMM_FORCE"); f_bn = 0; if (e) sscanf(e, "%d:%d:%d", &f_bn, &f_s, &f_st); }
    if (f_bn == 256 && batch == 1) return f_st == 2 ? launch_cfg<256, 128, 4, 2, 2>(p, 1, st) : launch_cfg<256, 128, 4, 2, 3>(p, 1, st);
    if (f_bn == 128 && p.geglu && batch == 1) return f_st == 4 ? launch_cfg<128, 128, 2, 2, 4>(p, 1, st) : launch_cfg<128, 128, 2, 2>(p, 1, st);
    if (f_bn > 0 && !p.geglu && batch == 1) {
      int S = f_s; if (S > nk / 2) S = nk / 2; if (S < 1) S = 1;
      if (S >= 2) CK0(ensure_splitk(p, S));
      if (f_bn == 64) return f_st == 4 ? launch_cfg<64, 64, 2, 2, 4>(p, S, st) : launch_cfg<64, 64, 2, 2>(p, S, st);
      if (f_bn == 1064) return f_st == 4 ? launch_cfg<64, 160, 2, 2, 4>(p, S, st) : launch_cfg<64, 160, 2, 2>(p, S, st);
      if (f_bn == 2064) return f_st == 4 ? launch_cfg<64, 128, 2, 2, 4>(p, S, st) : launch_cfg<64, 128, 2, 2>(p, S, st);
      if (f_bn == 160) return f_st == 4 ? launch_cfg<128, 160, 2, 2, 4>(p, S, st) : launch_cfg<128, 160, 2, 2>(p, S, st);
      return f_st == 4 ? launch_cfg<128, 128, 2, 2, 4>(p, S, st) : launch_cfg<128, 128, 2, 2>(p, S, st);
    }
  }
#endif
  // 3x3 stride-1 convs of the 16 x 16 / 8 x 8 maps through the producer / consumer kernel (igemm_pc.h; IgemmP::pc bit 1: 256 unsplit tiles of 64 x 160, bit 2: 128 tiles of
  // 128 x 160 x 2 K slices, bit 3: the 8 x 8 maps' 64 tiles of 64 x 160 x 4 K slices).  The loaders compute the im2col offsets of every (tap, chunk) step.
  if ((p.pc & 14) && p.ksize == 3 && pc_ok(p) && (p.N % 160) == 0 && (p.M % 64) == 0 && !p.w_per_image && batch == 1) {
    const long long t64 = (long long)(p.M / 64) * (p.N / 160);
    if ((p.pc & 2) && t64 >= 192 && t64 <= 256) return launch_pc<64, 160, 4, 5>(p, 1, st);
    if ((p.pc & 4) && !nosplit && (p.M % 128) == 0 && t64 >= 192 && t64 <= 256 && nk >= 32) { CK0(ensure_splitk(p, 2)); return launch_pc<128, 160, 4, 4>(p, 2, st); }
    if ((p.pc & 8) && !nosplit && t64 <= 64 && nk >= 64) {
      int S = (int)(256 / t64); if (S > 8) S = 8; if (S > nk / 16) S = nk / 16;
      if (S >= 2) { CK0(ensure_splitk(p, S)); return launch_pc<64, 160, 4, 5>(p, S, st); }
    }
  }
  // 8 x 8 maps, 3x3 stride 1: whole images resident, every weight tile streamed once (igemm_smap.h)
  if (p.smap && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.up == 1 && p.Hin == 8 && p.Win == 8 && p.Hout == 8 && p.Wout == 8 && batch == 1 && !p.geglu &&
      !p.w_per_image && (p.N % 64) == 0 && (p.M % 64) == 0 && !nosplit_early(p) && (long long)p.N * p.K * 2 < (1LL << 32)) {
    const int tiles = ((p.M + 511) / 512) * (p.N / 64), nch = (p.C0 + p.C1) >> 6;
    int smax = 256 / tiles; if (smax < 1) smax = 1; if (smax > nch) smax = nch;
    const int per = (nch + smax - 1) / smax, S = (nch + per - 1) / per;
    if (p.cfg_out) { p.cfg_out[0] = 512; p.cfg_out[1] = 64; p.cfg_out[2] = S; return 0; }
#ifdef AGD_EXPERIMENTS
    { static const bool logit = getenv("AGD_IGEMM_LOG") != nullptr;
      if (logit) fprintf(stderr, "IGEMM M=%d N=%d K=%d ks=3 stride=1 up=1 geglu=0 res=%d tile=512x64 splits=%d batch=1\n", p.M, p.N, p.K, p.residual ? 1 : 0, S); }
#endif
    constexpr int lds = 800 * 128 + 6 * 64 * 128 + 8192;
    if (S >= 2) CK0(ensure_splitk(p, S));
    const void* kfn = S >= 2 ? (const void*)igemm_smap_kernel<1> : (const void*)igemm_smap_kernel<0>;
    static std::atomic<bool> attr[AGD_MAX_DEVICES][2] = {};
    int dev = 0; HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= AGD_MAX_DEVICES) { agd_set_error("igemm_smap: device ordinal %d out of range", dev); return -1; }
    if (!attr[dev][S >= 2]) { HIP_CHECK_RET(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr[dev][S >= 2] = true; }
    IgemmP pp = p;
    void* args[] = {&pp};
    HIP_CHECK_RET(hipLaunchKernel(kfn, dim3((unsigned)tiles, 1, (unsigned)S), dim3(512), args, lds, st));
    if (S >= 2) return launch_splitk_reduce(pp, S, st);
    return 0;
  }
  // weight-streaming kernel (igemm_wreg.h): plain 1x1 launches of the small maps whose matrix exists in fragment order
  if (p.Wfrag && p.wreg && p.ksize == 1 && batch == 1 && !p.w_per_image && p.C1 == 0 && p.C0 == p.K && p.stride == 1 && p.up == 1 && p.pad == 0 &&
      p.Hin == p.Hout && p.Win == p.Wout && !p.out_f32 && p.M >= p.wreg_mmin && p.M <= p.wreg_mmax &&
      (p.geglu ? (p.wfrag_ni == 4 && (p.wreg & 2) && p.N % 256 == 0) : (p.wfrag_ni == 2 && (p.wreg & 1) && p.N % 128 == 0))) {
    // plain launches: 160-wide tiles (five waves) where they give a fuller single wave of workgroups than the 128-wide ones
    const long long mt64 = (p.M + 63) / 64;
    const bool w160 = !p.geglu && p.N % 160 == 0 && mt64 * (p.N / 160) <= 256 && mt64 * (p.N / 128) > 256;
    const int bn = p.geglu ? 256 : w160 ? 160 : 128;
    const long long tiles = mt64 * (p.N / bn);
    if (tiles >= 128) {
      if (p.cfg_out) { p.cfg_out[0] = 64; p.cfg_out[1] = bn; p.cfg_out[2] = 1; return 0; }
      if (p.colstat_out && (p.colstat_rows < 1 || p.colstat_rows % 64)) { agd_set_error("igemm_wreg: column statistics need M tiles inside one image"); return -1; }
      if (p.rowstat_out && p.rowstat_slots != p.N / bn) { agd_set_error("igemm_wreg: rowstat_slots %d != N tiles %d", p.rowstat_slots, p.N / bn); return -1; }
#ifdef AGD_EXPERIMENTS
      { static const bool logit = getenv("AGD_IGEMM_LOG") != nullptr;
        if (logit) fprintf(stderr, "IGEMM M=%d N=%d K=%d ks=1 stride=1 up=1 geglu=%d res=%d tile=64x%d splits=1 batch=1\n", p.M, p.N, p.K, p.geglu, p.residual ? 1 : 0, bn); }
#endif
      // bit 2 of p.wreg: two K groups of waves per workgroup on the plain launches of at most one workgroup per CU (an even number of 64-deep stages each)
      const bool kg2 = (p.wreg & 4) && !p.geglu && tiles <= 256 && ((p.K >> 6) & 1) == 0 && (p.K >> 6) >= 8;
      const void* kfn = p.geglu ? (const void*)igemm_wreg_kernel<4, 1> :
                        w160 ? (kg2 ? (const void*)igemm_wreg_kernel<2, 0, 5, 2> : (const void*)igemm_wreg_kernel<2, 0, 5>) :
                               (kg2 ? (const void*)igemm_wreg_kernel<2, 0, 4, 2> : (const void*)igemm_wreg_kernel<2, 0>);
      const int wn = w160 ? 5 : 4, kgn = kg2 ? 2 : 1;
      IgemmP pp = p; pick_xcd_block(pp, 64, bn);
      void* args[] = {&pp};
      HIP_CHECK_RET(hipLaunchKernel(kfn, dim3((unsigned)tiles), dim3(wn * 64 * kgn), args, kgn * (3 * 64 * 128 + wn * 1024) + 64 * 8, st));
      return 0;
    }
  }
  if (p.p8) { const int c8 = pick_8p(p); if (c8 == 1) return launch_8p<2, 4, 8, 2, 2>(p, st); if (c8 == 2) return launch_8p<4, 2, 4, 3, 2>(p, st); }
  if (p.geglu) return launch_cfg<128, 128, 2, 2>(p, 1, st);
  // 1x1 launches of the 16x16 / 8x8 maps with N a multiple of 160: 64 x 160 tiles on the 4-stage ring give exactly (or, with K slices,
  // up to) 256 workgroups where the 128-row tiles give 160 or need a K split -- M = 2048, N = 1280 (tools/kb_force64.py, hot operands):
  // K = 1280 17.5 -> 16.1 us, K = 2560 25.4 -> 23.3, K = 5120 45.5 (128 x 160, 2 slices) -> 39.4 unsplit; M = 512, K = 5120: 26.6 (8 slices) ->
  // 21.9 (4 slices).  In situ (tools/ab_bench_libs.sh): 548.3 -> 539.8 ms per batch.  3x3 launches and every other shape tried are better off on
  // the 128-row tiles; in situ also: 512 tiles of 64 x 160 for the 32x32 maps' C->C launches (+0.5 .. 8 ms) and 64 x 128 GEGLU tiles for M <= 2048 (+0.3 ms).
  if (!KNOB(14) && batch == 1 && p.ksize == 1 && (p.N % 160) == 0 && (p.M % 64) == 0 && nk >= 16) {
    const long long t64 = (long long)(p.M / 64) * (p.N / 160);
    // (two K groups of waves on these tiles: 14.3 -> 17.5 us at M = 2048, K = N = 1280 -- each wave issues 7 LDS-DMA pieces per 20 MFMAs and that issue
    //  cost, not latency, is what a K step waits for; a second group doubles it.  tools/kb_m512.py)
    if (t64 >= 192 && t64 <= 256) {
      // producer / consumer form (igemm_pc.h): loader waves issue the ring's LDS-DMA pieces back to back, consumer waves only read fragments and run MFMAs
      if ((p.pc & 1) && pc_ok(p)) return launch_pc<64, 160, 4, 5>(p, 1, st);
      return launch_cfg<64, 160, 2, 2, 4>(p, 1, st);
    }
    if (t64 <= 64 && nk >= 64 && !nosplit) {
      int S = (int)(256 / t64); if (S > nk / 16) S = nk / 16;
      if (S >= 2) { CK0(ensure_splitk(p, S)); return launch_cfg<64, 160, 2, 2, 4>(p, S, st); }
    }
  }
  // Small-M launches (8x8 / 16x16 feature maps): ONE workgroup per CU on the 4-stage ring, tile width and K split
  // chosen so that the launch has as close to 256 workgroups as possible.  These launches are a load-latency chain:
  // two more stages in flight hide more of it than a second co-resident workgroup on a 2-stage ring does
  // (8x8 1280->1280: 35 -> 30 us).  16x16 maps (M = 2048, N = 1280) get 128 tiles of 128x160 x 2 K slices = 256.
  if (!KNOB(6) && batch == 1 && t128 <= 160 && nk >= 64 && p.M >= 128 && (p.N & 3) == 0) {
    auto splits = [&](long long tt) { if (nosplit) return 1; int S = (int)(256 / tt); if (S > 8) S = 8; if (S > nk / 8) S = nk / 8; return S < 1 ? 1 : S; };
    const long long mt = (p.M + 127) / 128;
    const int S8 = splits(t128);
    long long T8 = t128 * S8, T0 = 0; int S0 = 1;
    if ((p.N % 160) == 0) { const long long t160 = mt * (p.N / 160); if (t160 <= 256) { S0 = splits(t160); T0 = t160 * S0; } }
    const bool use160 = T0 > T8 && !KNOB(7);
    const int S = use160 ? S0 : S8;
    const long long T = use160 ? T0 : T8;
    if (T >= 192 || t128 <= 64) {                                   // else: too few workgroups -> 2-stage path below
      if (S >= 2) CK0(ensure_splitk(p, S));
      return use160 ? launch_cfg<128, 160, 2, 2, 4>(p, S, st) : launch_cfg<128, 128, 2, 2, 4>(p, S, st);
    }
  }
  // split-K for small-M problems (8x8 / 16x16 feature maps): fill the 256 CUs with K slices
  if (batch == 1 && t128 <= 160 && nk >= 64 && (p.N & 3) == 0 && p.M >= 128 && !nosplit) {
    int S = (int)((384 + t128 - 1) / t128);
    if (S > 8) S = 8;
    if (S > nk / 8) S = nk / 8;
    if (S >= 2) {
      CK0(ensure_splitk(p, S));
      return launch_cfg<128, 128, 2, 2>(p, S, st);
    }
  }
  // N <= 64 (conv_out: 4 / 3 output channels): a 128-wide tile would be > 95 % padding -> 64x64 tiles
  if (p.N <= 64) return launch_cfg<64, 64, 2, 2, 4>(p, 1, st);
  if (t128 >= 192) {
    // 128x128 or 128x160: both hold 2 workgroups per CU.  Wave quantisation decides: a launch of T tiles keeps
    // every CU busy for ceil(T/256) tile-times (x1.33 when T <= 256: a lone workgroup per CU has nothing to overlap
    // its loads/epilogue with), and a 160-wide tile is 1.25 tile-times.  E.g. M=8192, N=640: 320 tiles of 128x128
    // leave 192 CUs idle for the second half; 256 tiles of 128x160 do not.
    bool n160 = (p.N % 160) == 0 && (p.N % 128) != 0;
    if ((p.N % 160) == 0 && (p.N % 128) == 0) {
      const long long mt = (p.M + 127) / 128;
      const long long T8 = mt * (p.N / 128) * batch, T0 = mt * (p.N / 160) * batch;
      auto cost = [](long long T, double w) { return w * (T <= 256 ? 1.33 : (double)((T + 255) / 256)); };
      // measured: helps 3x3 and long-K 1x1 (M=8192 K=2560 N=640: 39.3 -> 34.0 us as 256 tiles on the deep ring), neutral for short-K 1x1
      if (p.ksize == 3 || nk >= 32 || KNOB(10)) n160 = cost(T0, 1.25) < cost(T8, 1.0) - 1e-9;
    }
    // <= 256 tiles: one workgroup per CU whatever the ring -> take the 4-stage ring (147 / 128 KB LDS)
    const long long Tsel = (long long)((p.M + 127) / 128) * ((p.N + (n160 ? 159 : 127)) / (n160 ? 160 : 128)) * batch;
    const bool deep = Tsel <= 256 && nk >= 8 && !KNOB(9);
    // at most HALF a round of 128 x 160 tiles (the stride-2 downsampling conv of the 64 x 64 maps: M = 8192, N = 320 -> 128 tiles): 64-row tiles fill the chip instead
    if (n160 && deep && !KNOB(12) && Tsel <= 128 && batch == 1 && !halo_ok(p) && (p.M % 64) == 0 && (long long)(p.M / 64) * (p.N / 160) <= 256) return launch_cfg<64, 160, 2, 2, 4>(p, 1, st);
    // (IgemmP::pc bit 5: the 1x1 launches of one 128 x 160 tile per CU on the producer / consumer kernel)
    if (n160 && deep && (p.pc & 32) && p.ksize == 1 && batch == 1 && !p.geglu && pc_ok(p)) return launch_pc<128, 160, 4, 4>(p, 1, st);
    if (n160) return deep ? launch_cfg<128, 160, 2, 2, 4>(p, 1, st) : launch_cfg<128, 160, 2, 2>(p, 1, st);
    return deep ? launch_cfg<128, 128, 2, 2, 4>(p, 1, st) : launch_cfg<128, 128, 2, 2>(p, 1, st);
  }
  // 128..191 tiles of 128x128 with a long K (M=2048 K=1280 N=1280: 160 tiles): one workgroup per CU on the deep ring beats 640
  // tiles of 64x64, whose LDS traffic per MFMA is twice as high (19.7 -> 16.5 us)
  if (t128 >= 128 && nk >= 16 && !KNOB(13)) return launch_cfg<128, 128, 2, 2, 4>(p, 1, st);
  if (p.kg2 && p.ksize == 1 && nk >= 8 && (long long)((p.M + 63) / 64) * ((p.N + 63) / 64) <= 256) return launch_cfg<64, 64, 2, 2, 4, 2, 4>(p, 1, st);
  return launch_cfg<64, 64, 2, 2, 4>(p, 1, st);
}

// can this 3x3 launch take its block's 1x1 conv_shortcut as extra K (IgemmP::sc0)?  True when the launcher picks the row-halo kernel for it.
bool igemm_can_fuse_shortcut(const IgemmP& p_in) {
  IgemmP p = p_in;                       // WITH the shortcut fields and the widened K: the launcher's decision for exactly the launch that would follow
  int cfg[3] = {0, 0, 0};
  p.cfg_out = cfg;
  if (!p.sc0 || p.ksize != 3) return false;
  const bool smap8 = p.smap && p.stride == 1 && p.pad == 1 && p.up == 1 && p.Hin == 8 && p.Win == 8 && p.Hout == 8 && p.Wout == 8;
  if (!halo_ok(p) && !smap8) return false;
  if (launch_igemm(p, nullptr) != 0) return false;
  return (cfg[0] == 128 && (cfg[1] == 128 || cfg[1] == 160) && halo_ok(p)) || (cfg[0] == 512 && cfg[1] == 64);      // the row-halo tiles (split-K or not), or the 8 x 8 whole-images kernel
}

// the tile configuration launch_igemm would pick for this problem: cfg3 = {BM, BN, K splits} (nothing is launched)
int igemm_query(const IgemmP& p_in, int* cfg3) {
  IgemmP p = p_in;
  p.cfg_out = cfg3;
  return launch_igemm(p, nullptr);
}
