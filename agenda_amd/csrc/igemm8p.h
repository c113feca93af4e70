// 256-row implicit GEMM on the 8-wave / 8-phase schedule (included by igemm.hip).
//
// Same contract as igemm_kernel (igemm.hip): out[M][N] = epilogue(sum_k A[m][k] W[n][k]) with A gathered on the fly from one or
// two NHWC sources (3x3 / 1x1 taps, stride, pad, nearest-2x upsample, channel concat), weights [N][K] K-contiguous, the epilogue of
// igemm_epilogue.h straight from the accumulator registers.  What differs is the main loop: the structure of
// cdna_hip_programming.md "The 256^2 8-phase template" (measured stand-alone in tools/gemm8p/gemm8p.hip) instead of the
// 4-wave / one-barrier-per-K-step ring, for launches with enough 256-row tiles to fill the chip.
//
//  * 512 threads = 8 waves as WM x WN; a wave owns MI x NI tiles of 16 x 16 (pixels x channels): 2 x 4 waves of 128 x 64 for the
//    256-wide tile, 4 x 2 waves of 64 x 80 for the 160-wide one (SD's widths are multiples of 320).
//  * One K tile (64 deep) lives in LDS as four parts: A half h = the pixel sub-blocks {wave row * WTM + h * WTM/2 ..} of all wave
//    rows (128 rows of 128 B), B part 0 / 1 = the first NI0 / last NI1 16-row MFMA tiles of every wave column.  Two K tiles are
//    resident.  LDS rows are in FRAGMENT order (the weight-row permutation of igemm_epilogue.h and the half / part split are
//    applied on the LDS-DMA source address), 16-B chunks XOR-swizzled with (row & 7): conflict-free ds_read_b128.
//  * A phase = {fragment reads of one part | LDS-DMA of one part of a later K tile} -> s_barrier -> MFMAs of one accumulator
//    quadrant over the whole K tile -> s_barrier.  Four phases per K tile:
//        P1 reads W part 0 + X half 0, stages A half 1 of tile t+1     -> acc[X0][W0]
//        P2 reads W part 1,            stages B part 0 of tile t+2     -> acc[X0][W1]
//        P3 reads X half 1,            stages A half 0 of tile t+2     -> acc[X1][W0]
//        P4 reads nothing,             stages B part 1 of tile t+2, counted vmcnt -> acc[X1][W1]
//    Waves 4..7 run the same program ONE BARRIER behind waves 0..3, so on every SIMD one wave computes while its partner reads
//    and stages.  vmcnt is waited for once per K tile (P4), counted: the three parts issued last stay in flight, everything older
//    (= all of tile t+1) has landed; tile t+1 is first read in the NEXT phase (behind P4's second barrier, which also orders the
//    other wave group's pieces).  A part is restaged two phases after its last read, or one phase after when a counted lgkmcnt
//    ahead of the reading phase's first barrier retired those reads (P1's W reads).
#pragma once
#include "kernels.h"
#include "igemm_epilogue.h"
#include <type_traits>

template <int N> AGD_DEV void p8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
template <int N> AGD_DEV void p8_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory"); }
#define P8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int WM, int WN, int MI, int NI0, int NI1>
struct P8Geom {
  static constexpr int NI = NI0 + NI1, BM = WM * MI * 16, BN = WN * NI * 16, WTM = MI * 16, WTN = NI * 16, HM = WTM / 2;
  static constexpr int A_HALF = 128 * 128;                                  // bytes: BM / 2 rows of 128 B
  static constexpr int B0_ROWS = WN * NI0 * 16, B1_ROWS = WN * NI1 * 16;
  static constexpr int B0_OFF = 2 * A_HALF, B1_OFF = B0_OFF + B0_ROWS * 128, KT = B1_OFF + B1_ROWS * 128;   // one K tile
  static constexpr int NP0 = B0_ROWS / 8, NP1 = B1_ROWS / 8;                // LDS-DMA pieces (8 rows each) per B part
  static constexpr int IT0 = (NP0 + 7) / 8, IT1 = (NP1 + 7) / 8;            // piece slots per wave (piece = wid + 8 i, live while < NP)
  static constexpr int RING = 2 * KT;
  static constexpr int EPI = BM * BN * 2 + WM * BN * 8;                     // epilogue staging (GroupNorm statistics: bf16 tile + column sums)
  static constexpr int STATS_OFF = (RING > EPI ? RING : EPI);                // LayerNorm-fold consumers: (mean, rstd) of the tile's 256 rows, 2 KiB
  static constexpr int LDS = STATS_OFF + 2048 + 8192;                        // + one scratch KiB per wave (cold-weight warm-up pieces)
  static_assert(WM * WN == 8 && BM == 256 && (MI % 2) == 0, "8 waves, 256-row tile");
  static_assert((NP0 % 8 == 0 || NP0 % 8 == 4) && (NP1 % 8 == 0 || NP1 % 8 == 4), "piece counts must be uniform per wave group");
};

template <int WM, int WN, int MI, int NI0, int NI1, int KS, int GEGLU>
__global__ __launch_bounds__(512, 2) void igemm8p_kernel(const IgemmP p) {
  using G = P8Geom<WM, WN, MI, NI0, NI1>;
  constexpr int NI = G::NI, BM = G::BM, BN = G::BN, WTM = G::WTM, WTN = G::WTN, HM = G::HM, KT = G::KT;
  constexpr int MH = MI / 2;
  constexpr unsigned OOB = 0x80000000u, LIVE = 0x7FFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef AGD_EXPERIMENTS
  if (p.dbg & 32) return;                          // timing experiment (agd_set_igemm_cfg(512)): dispatch cost of this grid only
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AGD_STAMPS
  // time stamps (tools/kb_8p_trace.py): dbg bit 6; bit 7 picks wave 4 (the group that runs one barrier behind) instead of wave 0
  const bool ts_on = (p.dbg & 64) && tid == ((p.dbg & 128) ? 256 : 0) && (int)blockIdx.x == g_smap_ts_wg;
  int ts_n = 0;
  if (ts_on) g_smap_ts[1020] = __builtin_amdgcn_s_memrealtime();
#endif
  AGD_TS(1);
  const int grp = wid >> 2;                           // waves 4..7 share the SIMDs of waves 0..3: they run one barrier behind
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tn, tm;
  tile_of(bid, (p.M + BM - 1) / BM, tiles_n, p.wmajor, p.xb_m, p.xb_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = p.K >> 6;
  const bf16_t* const base0 = p.src0;
  const bf16_t* const base1 = p.src1 ? p.src1 : p.src0;

  // ---- LDS-DMA sources.  A piece = one wave instruction = 8 LDS rows of 128 B; lane: row lrow of the piece, position lane & 7
  // holds the logical 16-B chunk (lane & 7) ^ (row & 7).  This wave issues pieces wid and wid + 8 of every A half.
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  unsigned bvoff0[G::IT0], bvoff1[G::IT1];
#pragma unroll
  for (int i = 0; i < G::IT0; ++i) {
    const int R = (wid + 8 * i) * 8 + lrow;                                // LDS row of part 0: wave column, MFMA tile jj, fragment row rho = 4q' + r'
    const int wco = R / (NI0 * 16), jj = (R >> 4) % NI0, rho = R & 15;
    const int n = n0 + wco * WTN + (rho >> 2) * (4 * NI) + 4 * jj + (rho & 3);
    bvoff0[i] = (R < G::B0_ROWS && n < p.N) ? (unsigned)(((long long)n * p.K + lchunk * 8) * 2) : OOB;
  }
#pragma unroll
  for (int i = 0; i < G::IT1; ++i) {
    const int R = (wid + 8 * i) * 8 + lrow;
    const int wco = R / (NI1 * 16), jj = (R >> 4) % NI1, rho = R & 15;
    const int n = n0 + wco * WTN + (rho >> 2) * (4 * NI) + 4 * (NI0 + jj) + (rho & 3);
    bvoff1[i] = (R < G::B1_ROWS && n < p.N) ? (unsigned)(((long long)n * p.K + lchunk * 8) * 2) : OOB;
  }

  // A rows of this lane: half h, slot i -> LDS row R = (wid + 8 i) * 8 + lrow of the half -> tile pixel (R / HM) * WTM + h * HM + R % HM.
  // Their im2col offsets are fixed within a (tap, source) segment and recomputed from the row index when the segment changes (nothing
  // but the four offsets lives across the loop).
  const bool lin = KS == 1;                           // the launcher routes only stride-1 / pad-0 1x1 launches here: im2col row = pixel
  // KS = 2: the phase convs of an upsampling conv (IgemmP::ups4, see igemm.hip): this tile's phase (a, b) = n0 / Cout pads (1 - a) rows above, (1 - b) columns left
  const int ph4 = (KS == 2 && p.ups4) ? n0 / p.ups4 : 0;
  const int pad_y = (KS == 2 && p.ups4) ? 1 - (ph4 >> 1) : p.pad, pad_x = (KS == 2 && p.ups4) ? 1 - (ph4 & 1) : p.pad;
  const int HWo = p.Hout * p.Wout;
  const int ush = (p.up == 2) ? 1 : 0;
  const float inv_hwo = 1.0f / (float)HWo, inv_wo = 1.0f / (float)p.Wout;
  unsigned avoff[2][2];
  int seg_left = 0, tap = 0, cursrc = 0;
  unsigned asoff = 0, bsoff = 0;
  auto set_segment = [&](int tap_, int src_) {
    tap = tap_; cursrc = src_;
    const int kh = (KS > 1) ? tap / KS : 0, kw = (KS > 1) ? tap - kh * KS : 0;
    const int Cs = cursrc ? p.C1 : p.C0;
    seg_left = Cs >> 6; asoff = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int R = (wid + 8 * i) * 8 + lrow;
        const int m = m0 + (R / HM) * WTM + h * HM + (R % HM);
        bool ok = m < p.M;
        int pix = m;
        if (!lin) {
          const int mm = ok ? m : 0;
          const int b = fast_udiv(mm, HWo, inv_hwo); const int rem = mm - b * HWo;
          const int oy = fast_udiv(rem, p.Wout, inv_wo), ox = rem - oy * p.Wout;
          const int iy = oy * p.stride - pad_y + kh, ix = ox * p.stride - pad_x + kw;
          ok = ok && (unsigned)iy < (unsigned)(p.Hin << ush) && (unsigned)ix < (unsigned)(p.Win << ush);
          pix = (b * p.Hin + (iy >> ush)) * p.Win + (ix >> ush);
        }
        avoff[h][i] = ok ? (unsigned)(((long long)pix * Cs + lchunk * 8) * 2) : OOB;
      }
  };
  auto advance = [&]() {                              // state of the NEXT K tile to stage (B0, A0, B1 now, A1 in the following P1)
    asoff += 128u; bsoff += 128u; --seg_left;
    if (seg_left == 0) { if (cursrc == 0 && p.C1 > 0) set_segment(tap, 1); else if (tap + 1 < KS * KS) set_segment(tap + 1, 0); }
  };
  auto stageA = [&](int buf, int h, bool live) {
    const unsigned so = __builtin_amdgcn_readfirstlane(asoff), nr = live ? LIVE : 0u;
    char* d = smem + buf * KT + h * G::A_HALF + wid * 1024;
    if (__builtin_amdgcn_readfirstlane(cursrc)) { bufdma16(base1, d, avoff[h][0], so, nr); bufdma16(base1, d + 8192, avoff[h][1], so, nr); }
    else { bufdma16(base0, d, avoff[h][0], so, nr); bufdma16(base0, d + 8192, avoff[h][1], so, nr); }
  };
  auto stageB0 = [&](int buf, bool live) {
    const unsigned so = __builtin_amdgcn_readfirstlane(bsoff), nr = live ? LIVE : 0u;
    char* d = smem + buf * KT + G::B0_OFF + wid * 1024;
#pragma unroll
    for (int i = 0; i < G::IT0; ++i)
      if ((i + 1) * 8 <= G::NP0 || grp == 0) bufdma16(p.W, d + i * 8192, bvoff0[i], so, nr);   // a part of 8 i + 4 pieces: slot i only in waves 0..3
  };
  auto stageB1 = [&](int buf, bool live) {
    const unsigned so = __builtin_amdgcn_readfirstlane(bsoff), nr = live ? LIVE : 0u;
    char* d = smem + buf * KT + G::B1_OFF + wid * 1024;
#pragma unroll
    for (int i = 0; i < G::IT1; ++i)
      if ((i + 1) * 8 <= G::NP1 || grp == 0) bufdma16(p.W, d + i * 8192, bvoff1[i], so, nr);
  };

  // cold-weight warm-up (see igemm.hip): the launch's first workgroups stream W once through the caches, 1/nb each
  if (p.warm == 2) {
    const int tot = gridDim.x, nb = tot < 512 ? tot : 512;
    if ((int)blockIdx.x < nb) {
      const long long pieces = ((long long)p.N * p.K * 2) >> 10;
      const long long p0 = pieces * blockIdx.x / nb, p1 = pieces * (blockIdx.x + 1) / nb;
      char* wl = smem + G::LDS - 8192 + wid * 1024;
      for (long long pc = p0 + wid; pc < p1; pc += 8) bufdma16(p.W, wl, (unsigned)(pc * 1024 + lane * 16), 0u);
    }
  }

  // ---- fragment read addresses (16x16x32 operand: lane (q, rho) reads fragment row rho, k chunk kk*4 + q; key = rho & 7)
  const int q = lane >> 4, rho = lane & 15;
  const int sw0 = (q ^ (rho & 7)) << 4;
  const int xoff = (wm * HM + rho) * 128 + sw0;                       // + ii * 2048 ; kk = 1: ^ 64
  const int w0off = G::B0_OFF + (wn * NI0 * 16 + rho) * 128 + sw0;    // + jj * 2048
  const int w1off = G::B1_OFF + (wn * NI1 * 16 + rho) * 128 + sw0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[MH][2], wf[NI][2];

  auto readW0 = [&](int buf) {
    const char* s = smem + buf * KT;
#pragma unroll
    for (int jj = 0; jj < NI0; ++jj) { wf[jj][0] = *(const bf16x8*)(s + jj * 2048 + w0off); wf[jj][1] = *(const bf16x8*)(s + jj * 2048 + (w0off ^ 64)); }
  };
  auto readW1 = [&](int buf) {
    const char* s = smem + buf * KT;
#pragma unroll
    for (int jj = 0; jj < NI1; ++jj) { wf[NI0 + jj][0] = *(const bf16x8*)(s + jj * 2048 + w1off); wf[NI0 + jj][1] = *(const bf16x8*)(s + jj * 2048 + (w1off ^ 64)); }
  };
  auto readX = [&](int buf, int h) {
    const char* s = smem + buf * KT + h * G::A_HALF;
#pragma unroll
    for (int ii = 0; ii < MH; ++ii) { xf[ii][0] = *(const bf16x8*)(s + ii * 2048 + xoff); xf[ii][1] = *(const bf16x8*)(s + ii * 2048 + (xoff ^ 64)); }
  };
  auto mma = [&](auto mh_, auto np_) {
    constexpr int mh = decltype(mh_)::value, np = decltype(np_)::value;
    constexpr int J0 = np ? NI0 : 0, NJ = np ? NI1 : NI0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int ii = 0; ii < MH; ++ii)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
          acc[mh * MH + ii][J0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[J0 + jj][kk], xf[ii][kk], acc[mh * MH + ii][J0 + jj], 0, 0, 0);   // D = W . X^T
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  // pieces this wave leaves in flight across P4's wait: B part 0, A half 0, B part 1 of tile t + 2
  constexpr int INFL_G0 = G::IT0 + 2 + G::IT1;
  constexpr int INFL_G1 = G::NP0 / 8 + 2 + G::NP1 / 8;

  // one K tile (tile t, resident in buffer b)
  auto ktile = [&](auto b_, int t) {
    constexpr int b = decltype(b_)::value, o = b ^ 1;
    readW0(b); __builtin_amdgcn_sched_barrier(0); readX(b, 0);
    stageA(o, 1, t + 1 < nk);
    p8_wait_lgkm<2 * MH>();                         // the W reads (issued first) have returned: B part 0 of this buffer may be restaged in P2
    P8_BAR(); p8_wait_lgkm<0>(); __builtin_amdgcn_sched_barrier(0);
    mma(I0{}, I0{});
    P8_BAR();
    const bool live = t + 2 < nk;
    if (live) advance();
    readW1(b);
    stageB0(b, live);
    P8_BAR(); p8_wait_lgkm<0>(); __builtin_amdgcn_sched_barrier(0);
    mma(I0{}, I1{});
    P8_BAR();
    readX(b, 1);
    stageA(b, 0, live);
    P8_BAR(); p8_wait_lgkm<0>(); __builtin_amdgcn_sched_barrier(0);
    mma(I1{}, I0{});
    P8_BAR();
    stageB1(b, live);
    if (grp == 0) p8_wait_vm<INFL_G0>(); else p8_wait_vm<INFL_G1>();
    P8_BAR();
    mma(I1{}, I1{});
    P8_BAR();
  };

  // prologue: tile 0 (four parts), then B0, A0, B1 of tile 1
  set_segment(0, 0);
  stageB0(0, true); stageA(0, 0, true); stageB1(0, true); stageA(0, 1, true);
  const bool live1 = nk > 1;
  if (live1) advance();
  stageB0(1, live1); stageA(1, 0, live1); stageB1(1, live1);
  AGD_TS(2);
  // LayerNorm-fold consumer: one thread per tile row sums the producer's partial sums (its `slots` dependent loads run beside the
  // prologue's LDS-DMA) and leaves (mean, rstd) in LDS for the epilogue -- the eight rows of a lane would otherwise cost the epilogue
  // eight x slots exposed L2 round trips with nothing to overlap them (one workgroup per CU)
  if (p.ln_stats && tid < BM) {
    const int m = m0 + tid;
    float S = 0.f, Q = 0.f;
    if (m < p.M) for (int k = 0; k < p.ln_slots; ++k) { const f32x2 v = *(const f32x2*)(p.ln_stats + ((long long)m * p.ln_slots + k) * 2); S += v[0]; Q += v[1]; }
    const float mu = S * p.ln_invC;
    float var = Q * p.ln_invC - mu * mu; var = var < 0.f ? 0.f : var;
    *(f32x2*)(smem + G::STATS_OFF + tid * 8) = f32x2{mu, rsqrtf(var + p.ln_eps)};
  }
  if (grp == 0) p8_wait_vm<INFL_G0>(); else p8_wait_vm<INFL_G1>();
  P8_BAR();
  if (grp == 1) P8_BAR();
  AGD_TS(3);
  int t = 0;
  for (; t + 1 < nk; t += 2) { ktile(I0{}, t); AGD_TS(4); ktile(I1{}, t + 1); AGD_TS(4); }
  if (t < nk) ktile(I0{}, t);
  AGD_TS(5);
  if (grp == 0) P8_BAR();
  p8_wait_vm<0>();                                   // dead tail pieces still write zeros into LDS: let them land before the epilogue reuses it

  AGD_TS(6);
  igemm_epilogue<BM, BN, WM, WN, GEGLU, 0, 1>(p, acc, smem, lane, wm, wn, m0, n0, tn, 0, p.ln_stats ? (const float*)(smem + G::STATS_OFF) : nullptr);
#ifdef AGD_STAMPS
  if (ts_on) { AGD_TS(7); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); AGD_TS(8); g_smap_ts[1023] = ts_n; g_smap_ts[1021] = __builtin_amdgcn_s_memrealtime(); }
#endif
}
