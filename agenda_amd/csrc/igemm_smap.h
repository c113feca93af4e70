// 3x3 / stride 1 / pad 1 implicit GEMM for the 8 x 8 feature maps: whole images resident, every weight byte streamed exactly once.
// (included by igemm.hip; round 4)
//
// At 8 x 8 (UNet batch 8: M = 512 rows) the ResnetBlock2D convs are weight-streaming problems: 29.5 MB (1280 -> 1280) or 59 MB of bf16
// weights against 1.3 MB of activations.  igemm_kernel tiles them 128 x 160 x 8 K-slices: every weight tile is fetched by four row
// tiles and every im2col row by eight column tiles -- 828 KB through the LDS-DMA path per workgroup for 47 MFLOP (36 us + an 8 us slab
// pass per launch, 0.15 of the binding roof).  Here a workgroup owns ALL 512 rows x 64 output channels x a slice of the input
// channels:
//   * the slice is walked in 64-channel chunks; ONE LDS image per chunk holds the eight images' pixels WITH their zero border
//     (8 x 10 x 10 halo rows of 128 B, written by LDS-DMA: border rows get an out-of-range offset and the hardware writes zeros), and
//     all nine taps read their MFMA fragments from it at a per-tap row offset -- 64 KB of activations per 9 x 64 k instead of 576 KB;
//   * each weight tile (64 rows x 64 k of one tap) is fetched once by exactly one workgroup of the launch: the matrix crosses
//     HBM -> L2 -> LDS once, through a 4-slot LDS-DMA ring;
//   * 8 waves = 4 row groups x 2 column groups, wave tile 128 rows (two images) x 32 columns; operand roles swapped as everywhere,
//     so igemm_epilogue.h runs behind it unchanged (split-K slabs; the slab pass -- fused with the GroupNorm that follows -- finishes).
// LDS bank conflicts: the swizzle key of a halo row is its x coordinate (& 7): with it the 16 rows of every fragment read (two image
// rows of eight pixels, any tap) fall on 16 distinct 16-byte slots (checked exhaustively for all tiles / taps / k-steps).
#pragma once
#include "kernels.h"
#include "igemm_epilogue.h"
#include <type_traits>

#ifdef AGD_STAMPS
// in-kernel time stamps of one wave (tools/kb_smap_trace.py): AGD_IGEMM_CFG bit 10 (dbg & 64); wave 0 of workgroup (g_smap_ts_wg, z = 0) stores s_memtime at every mark
#define SMAP_TS(k) AGD_TS(k)
#else
#define SMAP_TS(k) do { } while (0)
#endif

template <int SPLITK>
__global__ __launch_bounds__(512) void igemm_smap_kernel(const IgemmP p) {
  constexpr int BM = 512, BN = 64, WM = 4, WN = 2, NW = 8, WTN = 32, MI = 8, NI = 2;
  constexpr int HS = 8, HP = HS + 2, IMG = HP * HP;               // 8 x 8 maps, halo pitch 10, 100 halo rows per image
  constexpr int A_ROWS = 8 * IMG, A_BYTES = A_ROWS * 128;         // 800 rows = 100 pieces of 8 rows
  constexpr int A_IT = (A_ROWS / 8 + NW - 1) / NW;                // 13 pieces per wave (the last four waves' 13th piece is dead)
  constexpr int BST = 6, B_BYTES = BN * 128;                      // weight ring: 6 slots of [64 rows][64 k]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;
  char* const sBr = smem + A_BYTES;
  char* const scr = sBr + BST * B_BYTES;                          // 8 KiB: dead-piece sink (one KiB per wave)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AGD_STAMPS
  const bool ts_on = (p.dbg & 64) && tid == 0 && (int)blockIdx.x == g_smap_ts_wg && blockIdx.z == 0;
  int ts_n = 0;
#endif
  SMAP_TS(1);
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = p.N / BN;
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int img0 = m0 >> 6, nimg = (p.M >> 6) - img0;             // images of this tile (<= 8 are live)
  const int Ct = p.C0 + p.C1, c0n = p.C0 >> 6, nch = Ct >> 6;

  // chunk range of this K slice
  int ch0 = 0, ch1 = nch;
  if constexpr (SPLITK) {
    const int per = (nch + (int)gridDim.z - 1) / (int)gridDim.z;
    ch0 = (int)blockIdx.z * per; ch1 = ch0 + per < nch ? ch0 + per : nch;
    if (ch1 < ch0) ch1 = ch0;
  }

  // ---- A image fill: piece (i * 8 + wid) covers halo rows 8 piece .. + 7; this lane: row lane >> 3, 16-byte chunk lane & 7 ----
  const int lrow = lane >> 3;
  int a_pix[A_IT]; unsigned a_live = 0; int a_key[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int piece = i * NW + wid, hr = piece * 8 + lrow;
    const int im = hr / IMG, rem = hr - im * IMG, yy = rem / HP, xx = rem - yy * HP;
    const bool ok = piece < A_ROWS / 8 && im < nimg && yy >= 1 && yy <= HS && xx >= 1 && xx <= HS;
    a_pix[i] = ok ? ((img0 + im) * HS + (yy - 1)) * HS + (xx - 1) : 0;
    a_key[i] = xx & 7;
    a_live |= (ok ? 1u : 0u) << i;
  }
  auto a_issue = [&](int cc) {                                     // chunk cc of the concatenated channels
    const bool s1 = cc >= c0n;
    const int Cs = s1 ? p.C1 : p.C0;
    const bf16_t* base = s1 ? p.src1 : p.src0;
    const unsigned so = (unsigned)((s1 ? cc - c0n : cc) * 128);
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int piece = i * NW + wid;
      const unsigned voff = ((a_live >> i) & 1) ? (unsigned)(a_pix[i] * Cs + (((lane & 7) ^ a_key[i]) << 3)) * 2u : 0x80000000u;
      char* dst = piece < A_ROWS / 8 ? sA + piece * 1024 : scr + wid * 1024;
      bufdma16(base, dst, voff, so);
    }
  };

  // the first chunk's image goes out as soon as its offsets exist: the 100 KB fly under the rest of the prologue (weight offsets, accumulator and fragment set-up)
  // instead of behind it (stamps: 5.3 k cycles of prologue, then 3.1 k of fill, of a 42 k-cycle launch)
  if (ch1 > ch0) a_issue(ch0);

  // ---- weight ring: stage t = (chunk, tap): rows n0 .. n0 + 63, k = tap * Ct + chunk * 64 .. + 63; one piece per wave ----
  const int brow = wid * 8 + lrow;                                 // tile-local weight row this lane fetches
  const int bqp = (brow % WTN) / (4 * NI);
  const int bkey = (brow & 3) | ((bqp & 1) << 2);                  // permuted rows: fragment-row swizzle key (igemm.hip)
  const unsigned bvoff = (unsigned)(((long long)(n0 + brow) * p.K + (((lane & 7) ^ bkey) << 3)) * 2);
  const int nsteps = (ch1 - ch0) * 9;
  auto b_issue = [&](int t, bool live) {
    const int cc = ch0 + t / 9, tap = t - (t / 9) * 9;
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((tap * Ct + cc * 64) * 2));
    bufdma16(p.W, sBr + (t % BST) * B_BYTES + wid * 1024, bvoff, so, live ? 0x7FFFFFF0u : 0u);
  };

  if (nsteps > 0) {
#pragma unroll
    for (int s = 0; s < BST - 1; ++s) b_issue(s, s < nsteps);      // prologue: the first five weight stages
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses ----
  const int frow = lane & 15, q = lane >> 4;
  const int fx = frow & 7, fy = frow >> 3;                         // pixel of the row tile: (y0 + fy, fx); y0 = 2 (i & 3), image 2 wm + (i >> 2)
  const int abase = ((2 * wm) * IMG + fy * HP + fx) * 128;         // + (i >> 2) * IMG * 128 + 2 (i & 3) * HP * 128 + (ky * HP + kx) * 128
  int foffB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foffB[kk] = (frow >> 2) * (4 * NI * 128) + (frow & 3) * 128 + ((((kk << 2) + q) ^ (lane & 7)) << 4);

  // Main loop, software-pipelined over the two 32-deep halves of a tap (K step) and across taps: behind barrier B_t a wave reads the fragments of half kk1 of tap t UNDER
  // the MFMAs of half kk0 (read one half step earlier), then those of half kk0 of tap t + 1 under the MFMAs of kk1 -- one fragment read behind each of the first ten
  // MFMAs of a half, the other six cover the last reads' latency.  (Round 5, in-kernel time stamps: with barrier -> wait -> 10 reads -> 16 MFMAs -> 10 reads -> 16 MFMAs
  // a tap took ~2200 cycles for 1024 cycles of matrix-pipe work per SIMD -- the two waves of a SIMD run in lockstep and hide nothing for each other.)
  // Ring: stage t + 1 has landed before B_t (vmcnt(3): stages t + 2 .. t + 4 may fly), stage t + 5 goes out behind B_t into the slot of tap t - 1.  A chunk's first
  // tap has no predecessor in the same image: its kk0 fragments are read in the open behind the image's barrier.
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
  bf16x8 fa[2][MI], fb[2][NI];
  auto rd = [&](auto kk_tag, int aoff, int akey, const char* sB) __attribute__((always_inline)) {
    constexpr int kk = decltype(kk_tag)::value;
#pragma unroll
    for (int j = 0; j < NI; ++j) fb[kk][j] = *(const bf16x8*)(sB + j * 512 + foffB[kk]);
    const int ach = (((kk << 2) + q) ^ akey) << 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[kk][i] = *(const bf16x8*)(sA + aoff + ((i >> 2) * IMG + 2 * (i & 3) * HP) * 128 + ach);
  };
  auto mm = [&](auto kk_tag) __attribute__((always_inline)) {
    constexpr int kk = decltype(kk_tag)::value;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);   // D = W . X^T
  };
  auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < MI + NI; ++n) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI - (MI + NI), 0);
  };
  int t = 0, bs = 0;                                               // tap counter of this K slice; ring slot of tap t
  auto tapstep = [&](auto tap_tag) __attribute__((always_inline)) {
    constexpr int TAP = decltype(tap_tag)::value, ky = TAP / 3, kx = TAP - ky * 3, kyn = (TAP + 1) / 3, kxn = (TAP + 1) - kyn * 3;
    const int bsn = bs + 1 == BST ? 0 : bs + 1;
    const char* sB = sBr + bs * B_BYTES + wn * WTN * 128;
    SMAP_TS(2);
    if constexpr (TAP > 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(BST - 3) : "memory");
      SMAP_TS(5);
      asm volatile("s_barrier" ::: "memory");                      // B_t
      SMAP_TS(6);
    }
    b_issue(t + BST - 1, t + BST - 1 < nsteps);                    // into the slot tap t - 1 released
    __builtin_amdgcn_sched_barrier(0);
    rd(K1{}, abase + (ky * HP + kx) * 128, (fx + kx) & 7, sB);
    mm(K0{});
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAP < 8) {
      rd(K0{}, abase + (kyn * HP + kxn) * 128, (fx + kxn) & 7, sBr + bsn * B_BYTES + wn * WTN * 128);
      mm(K1{});
      interleave();
    } else {
      mm(K1{});
    }
    __builtin_amdgcn_sched_barrier(0);
    ++t; bs = bsn;
  };
  if (nsteps > 0) {
    for (int c = ch0; c < ch1; ++c) {
      // chunk boundary: every wave has left the previous chunk's last tap -> refill the image (the first one is already on its way), wait for everything in flight
      SMAP_TS(2);
      if (c > ch0) {
        asm volatile("s_barrier" ::: "memory");
        SMAP_TS(3);
        a_issue(c);
      }
      SMAP_TS(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SMAP_TS(5);
      asm volatile("s_barrier" ::: "memory");
      SMAP_TS(6);
      rd(K0{}, abase, fx & 7, sBr + bs * B_BYTES + wn * WTN * 128);
      __builtin_amdgcn_sched_barrier(0);
      tapstep(std::integral_constant<int, 0>{}); tapstep(std::integral_constant<int, 1>{}); tapstep(std::integral_constant<int, 2>{});
      tapstep(std::integral_constant<int, 3>{}); tapstep(std::integral_constant<int, 4>{}); tapstep(std::integral_constant<int, 5>{});
      tapstep(std::integral_constant<int, 6>{}); tapstep(std::integral_constant<int, 7>{}); tapstep(std::integral_constant<int, 8>{});
    }
  }
  SMAP_TS(8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // dead tail pieces still write zeros to LDS: let them land
  SMAP_TS(9);
  if (p.sc0) {
    // ---- the block's 1x1 conv_shortcut as extra K (IgemmP::sc0, as in igemm_halo.h): 64-channel chunks of the raw block input (one or two sources), dealt to the K
    // slices like the 3x3 chunks; per chunk one LDS image (the centre tap reads it) and ONE weight tile (columns 9 Ct + 64 r ... of the rows) into ring slot 0
    const int nsc = (p.sc_C0 + p.sc_C1) >> 6, sc0n = p.sc_C0 >> 6;
    int r0 = 0, r1 = nsc;
    if constexpr (SPLITK) { const int per = (nsc + (int)gridDim.z - 1) / (int)gridDim.z; r0 = (int)blockIdx.z * per; r1 = r0 + per < nsc ? r0 + per : nsc; if (r1 < r0) r1 = r0; }
    const int aoff = abase + (1 * HP + 1) * 128, akey = (fx + 1) & 7;
    for (int r = r0; r < r1; ++r) {
      asm volatile("s_barrier" ::: "memory");                      // every wave has left the previous image and weight slot 0
      const bool s1 = r >= sc0n;
      const int Cs = s1 ? p.sc_C1 : p.sc_C0;
      const bf16_t* base = s1 ? p.sc1 : p.sc0;
      const unsigned so = (unsigned)((s1 ? r - sc0n : r) * 128);
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int piece = i * NW + wid;
        const unsigned voff = ((a_live >> i) & 1) ? (unsigned)(a_pix[i] * Cs + (((lane & 7) ^ a_key[i]) << 3)) * 2u : 0x80000000u;
        char* dst = piece < A_ROWS / 8 ? sA + piece * 1024 : scr + wid * 1024;
        bufdma16(base, dst, voff, so);
      }
      bufdma16(p.W, sBr + wid * 1024, bvoff, __builtin_amdgcn_readfirstlane((unsigned)((9 * Ct + r * 64) * 2)));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      const char* sB = sBr + wn * WTN * 128;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 b[NI], a[MI];
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = *(const bf16x8*)(sB + j * 512 + foffB[kk]);
        const int ach = (((kk << 2) + q) ^ akey) << 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *(const bf16x8*)(sA + aoff + ((i >> 2) * IMG + 2 * (i & 3) * HP) * 128 + ach);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    }
    asm volatile("s_barrier" ::: "memory");                        // (the epilogue's LDS-staged paths start behind their own barriers; the register path touches no LDS)
  }
  igemm_epilogue<BM, BN, WM, WN, 0, SPLITK>(p, acc, smem, lane, wm, wn, m0, n0, tn, 0, nullptr);
#ifdef AGD_STAMPS
  if (ts_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SMAP_TS(10); g_smap_ts[1023] = ts_n; }
#endif
}
