// Producer / consumer ROW-HALO implicit GEMM: 3x3 / stride 1 / pad 1 convs on 128 x BN tiles at ONE workgroup per CU (the 32 x 32 maps' 256 tiles, the 16 x 16
// maps' 128 tiles x 2 K slices at UNet batch 8).  igemm_halo.h's operand scheme -- per (ky, source, 64-channel chunk) GROUP one LDS image of the tile's pixel rows
// with a halo pixel on either side, read by the three kx taps at row offsets 0 / 1 / 2; one weight tile per tap -- with igemm_pc.h's division of labour:
//   * 4 LOADER waves issue nothing but LDS-DMA pieces: 5 weight pieces each per K step (tap), 5 image pieces each per group; they compute the image's im2col
//     offsets themselves; rings: 3 A images, 5 weight stages (158 KB of LDS), counted vmcnt, the same count at every step;
//   * 4 CONSUMER waves (2 x 2 over the tile, wave tile 64 x BN / 2) read fragments and run MFMAs, software-pipelined over the two 32-deep halves of a K step:
//       barrier B_t | read kk1(t) | MFMA kk0(t) | read kk0(t + 1) | MFMA kk1(t)
//     so every batch of fragment reads lands under twenty MFMAs and only two fragment sets live in registers;
//   * ONE s_barrier per K step for all 8 waves.  Before B_t a loader has seen its share of weight stage t + 1 (and of the next group's image, when step t + 1 opens
//     a group) land; behind B_t it refills the weight slot of step t - 1 (stage t + 4) and, when step t opens a group, the image slot of the group before (image
//     group + 2).  Reads of step s are issued behind B_(s-1) and B_s and are complete when that step's last MFMAs issue, i.e. before B_(s+1): the slots are free.
// Why (DESIGN Appendix A, round 5): with one wave per SIMD doing both jobs the halo kernel's K step lasts ~0.67 us at M = 8192 for 0.30 us of matrix-pipe work: each
// LDS-DMA piece holds the issuing wave ~80 cycles, and the step is one basic block of barrier -> reads -> (MFMA, piece, read) x 9.  Same tiles, same fragment
// layout, same summation order as igemm_halo_kernel: bit-identical outputs.
#pragma once
#include "igemm_epilogue.h"
#include <type_traits>

template <int BN, int SPLITK>
__global__ __launch_bounds__(512) void igemm_pch_kernel(const IgemmP p, const int Wt, const int HRP) {
  constexpr int BM = 128, WM = 2, WN = 2, NW = 4, NLW = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int NA = 3, NB = 5;                          // ring depths: A images / weight stages
  constexpr int B_IT = BN * 8 / (NLW * 64), B_BYTES = BN * 128;
  constexpr int A_ITH = 5;                               // image pieces per loader wave (up to 160 halo rows)
  static_assert(B_IT == 5, "five weight pieces per loader wave and stage (BN = 160): one vmcnt for every step");
  constexpr int LPW = 5;
  const int A_BYTES = HRP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sAr = smem;                                // NA x A_BYTES
  char* const sBr = smem + NA * A_BYTES;                 // NB x B_BYTES
  char* const scr = sBr + NB * B_BYTES;                  // NLW KiB: dead-piece sink

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tn, tm;
  tile_of(bid, (p.M + BM - 1) / BM, tiles_n, 0, p.xb_m, p.xb_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int W = p.Wout, H = p.Hout, HW = H * W;
  const int hw2 = Wt + 2, trows = BM / Wt;
  const int Ct = p.C0 + p.C1;
  const int gpk = Ct >> 6, ngr = 3 * gpk;                // groups per ky; groups in all
  int g0 = 0, g1 = ngr;
  if constexpr (SPLITK) {
    const int per = (ngr + (int)gridDim.z - 1) / (int)gridDim.z;
    g0 = (int)blockIdx.z * per; g1 = g0 + per < ngr ? g0 + per : ngr;
    if (g1 < g0) g1 = g0;
  }
  const int nsteps = 3 * (g1 - g0);
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;

#ifdef AGD_STAMPS
  // time stamps (tools/kb_pch_trace.py): dbg bit 6; bit 7 picks loader wave 0 instead of consumer wave 0
  const bool ts_on = (p.dbg & 64) && (int)threadIdx.x == ((p.dbg & 128) ? NW * 64 : 0) && (int)blockIdx.x == g_smap_ts_wg && blockIdx.z == 0;
  int ts_n = 0;
  if (ts_on) g_smap_ts[1020] = __builtin_amdgcn_s_memrealtime();      // 100 MHz: the shader clock = s_memtime ticks / (s_memrealtime ticks x 10 ns)
#endif
  AGD_TS(1);
  if (wid >= NW) {
    // =========================== loader waves ===========================
    const int lw = wid - NW;
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;                // swizzle on the source side (LDS-DMA writes lane-linear)
    int a_brow[A_ITH], a_ix[A_ITH], a_y[A_ITH];
    unsigned a_ok = 0;
#pragma unroll
    for (int i = 0; i < A_ITH; ++i) {
      const int R = (i * NLW + lw) * 8 + lrow;           // halo row of the image this lane fetches
      const int j = R / hw2, col = R - j * hw2;
      const int pm = m0 + j * Wt;                        // first output pixel of tile row j
      bool ok = j < trows && pm < p.M;
      int b = 0, y = 0, x0 = 0;
      if (ok) { b = pm / HW; const int rem = pm - b * HW; y = rem / W; x0 = rem - y * W; }
      const int ix = x0 + col - 1;
      ok = ok && (unsigned)ix < (unsigned)W;
      a_brow[i] = b * p.Hin; a_ix[i] = ix; a_y[i] = y - 1;
      a_ok |= (ok ? 1u : 0u) << i;
    }
    unsigned bvoff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = (i * NLW + lw) * 8 + lrow, n = n0 + row;
      const int qp = (row % WTN) / (4 * NI);
      const int key = (row & 3) | ((qp & 1) << 2);       // permuted weight rows: the fragment row that reads this row, mod 8 (igemm_epilogue.h)
      bvoff[i] = (n < p.N) ? (unsigned)(((long long)n * p.K + ((lane & 7) ^ key) * 8) * 2) : 0x80000000u;
    }
    char* const sink = scr + lw * 1024;
    const int c0n = p.C0 >> 6;
    if (p.warm == 2) {                                   // cold-weight warm-up (igemm.hip): the first workgroups stream W once, 1 / nb each
      const int lin = blockIdx.z * gridDim.x + blockIdx.x;
      const int tot = gridDim.x * gridDim.z, nb = tot < 512 ? tot : 512;
      if (lin < nb) {
        const long long pieces = ((long long)p.N * p.K * 2) >> 10;
        const long long p0 = pieces * lin / nb, p1 = pieces * (lin + 1) / nb;
        for (long long pc = p0 + lw; pc < p1; pc += NLW) bufdma16(p.W, sink, (unsigned)(pc * 1024 + lane * 16), 0u);
      }
    }
    // image cursor: next group whose image goes out; weight cursor: next step (group, kx) whose stage goes out
    int ig = g0, ir = g0 / 3, iky = g0 - (g0 / 3) * 3;    // groups in igemm_halo.h's order: chunk outer, ky inner (A crosses L2 -> HBM once)
    int ws = 0, wky = iky, wr = ir, wkx = 0;
    // pieces [5 KX / 2 ...) of the image at the cursor: 2 + 2 + 1 over the three steps of a group (ten pieces behind one barrier held the loader ~1400 cycles against
    // the consumers' ~1000-cycle step: every third barrier waited for it); the cursor moves on behind the last part
    auto issue_image = [&](auto kx_tag) {
      constexpr int KXI = decltype(kx_tag)::value, I0 = KXI == 0 ? 0 : KXI == 1 ? 2 : 4, I1 = KXI == 0 ? 2 : KXI == 1 ? 4 : A_ITH;
      const bool live = ig < g1;
      const bool s1 = ir >= c0n;
      const int Cs = s1 ? p.C1 : p.C0;
      const bf16_t* abase = s1 ? p.src1 : p.src0;
      const unsigned aso = __builtin_amdgcn_readfirstlane((unsigned)((s1 ? ir - c0n : ir) * 128));
      char* const dA = sAr + ((ig - g0) % NA) * A_BYTES;
      const unsigned nr = live ? 0x7FFFFFF0u : 0u;
#pragma unroll
      for (int i = I0; i < I1; ++i) {
        const bool ok = ((a_ok >> i) & 1) && (unsigned)(a_y[i] + iky) < (unsigned)H;
        const int pix = (a_brow[i] + a_y[i] + iky) * p.Win + a_ix[i];
        const unsigned off = ok ? (unsigned)(pix * Cs + lchunk * 8) * 2u : 0x80000000u;            // < 2^31 bytes per source (launcher)
        const int pc = i * NLW + lw;
        bufdma16(abase, pc * 8 < HRP ? dA + pc * 1024 : sink, off, aso, nr);
      }
      if (KXI == 2 && live) { ++ig; if (++iky == 3) { iky = 0; ++ir; } }
    };
    auto issue_stage = [&]() {
      const bool live = ws < nsteps;
      const unsigned bso = __builtin_amdgcn_readfirstlane((unsigned)(((wky * 3 + wkx) * Ct + wr * 64) * 2));
      char* const dB = sBr + (ws % NB) * B_BYTES;
      const unsigned nr = live ? 0x7FFFFFF0u : 0u;
#pragma unroll
      for (int i = 0; i < B_IT; ++i) bufdma16(p.W, dB + (i * NLW + lw) * 1024, bvoff[i], bso, nr);
      ++ws;
      if (live && ++wkx == 3) { wkx = 0; if (++wky == 3) { wky = 0; ++wr; } }
    };
    // prologue: images of the first NA - 1 groups, weight stages 0 .. NB - 2
#pragma unroll
    for (int i = 0; i < NA - 1; ++i) { issue_image(K0{}); issue_image(K1{}); issue_image(K2{}); }
#pragma unroll
    for (int s = 0; s < NB - 1; ++s) issue_stage();
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NB - 2) * LPW) : "memory");       // image g0 and stage 0 have landed (this wave's pieces)
    asm volatile("s_barrier" ::: "memory");                                      // P
    // one step.  Issue order behind B_s is [stage s + 4, image part s]; before B_t stage t + 1 must have landed: at most the two youngest stages and the three youngest
    // image parts (2 + 2 + 1 pieces in any rotation) may be in flight: vmcnt(15).  When step t + 1 OPENS a group (kx = 2 steps), that group's image must have landed too, and
    // its last part (one piece, issued behind B_(t-3)) is YOUNGER than stage t + 1: it is the oldest of those fifteen, so these steps wait vmcnt(14) (with 15 the
    // consumers could read halo rows 128 .. 143 of the new image before they had arrived).  The first three steps have fewer image parts behind them (the images of groups
    // g0, g0 + 1 belong to the prologue and are older than every stage).
    auto lstep = [&](int t, auto kx_tag) {
      AGD_TS(2);
      if (t >= 3) { if constexpr (decltype(kx_tag)::value == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * LPW - 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * LPW) : "memory"); }
      else if (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * LPW) : "memory");
      else if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * LPW + 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * LPW + 4) : "memory");
      AGD_TS(3);
      asm volatile("s_barrier" ::: "memory");                                    // B_t
      AGD_TS(4);
      issue_stage();                                      // stage t + NB - 1 into the slot of stage t - 1
      issue_image(kx_tag);                                // a part of image group + NA - 1 into the slot of the group before this one (free since the group's first barrier)
    };
    for (int t = 0; t < nsteps; t += 3) { lstep(t, K0{}); lstep(t + 1, K1{}); lstep(t + 2, K2{}); }
    AGD_TS(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (dead tail pieces still write zeros into LDS)
    if (p.sc0) {
      // ---- the block's 1x1 conv_shortcut as extra K (IgemmP::sc0, as in igemm_halo.h): 64-channel chunks of the raw block input (one or two sources), the tile's own pixel
      // rows (ky = 1; the consumers read the centre tap), weight columns 9 Ct + 64 r ...; A images 0 / 1 and weight slots 0 / 1, chunk r + 1 requested behind barrier S_r
      // (the consumers have left chunk r - 1, whose slots it takes) and landed in full before S_(r + 1): it flies under the MFMAs of chunk r
      asm volatile("s_barrier" ::: "memory");             // Z: every consumer has left the 3x3 loop: the rings are free
      const int nsc = (p.sc_C0 + p.sc_C1) >> 6, sc0n = p.sc_C0 >> 6;
      int r0 = 0, r1 = nsc;                               // split-K: the chunks are dealt to the K slices like the 3x3 groups
      if constexpr (SPLITK) { const int per = (nsc + (int)gridDim.z - 1) / (int)gridDim.z; r0 = (int)blockIdx.z * per; r1 = r0 + per < nsc ? r0 + per : nsc; if (r1 < r0) r1 = r0; }
      auto issue_sc = [&](int r, bool live) {
        const bool s1 = r >= sc0n;
        const int Cs = s1 ? p.sc_C1 : p.sc_C0;
        const bf16_t* abase = s1 ? p.sc1 : p.sc0;
        const unsigned aso = __builtin_amdgcn_readfirstlane((unsigned)((s1 ? r - sc0n : r) * 128));
        const int slot = (r - r0) & 1;
        char* const dA = sAr + slot * A_BYTES;
        char* const dB = sBr + slot * B_BYTES;
        const unsigned nr = live ? 0x7FFFFFF0u : 0u;
#pragma unroll
        for (int i = 0; i < A_ITH; ++i) {
          const bool ok = ((a_ok >> i) & 1) && (unsigned)(a_y[i] + 1) < (unsigned)H;
          const int pix = (a_brow[i] + a_y[i] + 1) * p.Win + a_ix[i];
          const unsigned off = ok ? (unsigned)(pix * Cs + lchunk * 8) * 2u : 0x80000000u;
          const int pc = i * NLW + lw;
          bufdma16(abase, pc * 8 < HRP ? dA + pc * 1024 : sink, off, aso, nr);
        }
        const unsigned bso = __builtin_amdgcn_readfirstlane((unsigned)((9 * Ct + r * 64) * 2));
#pragma unroll
        for (int i = 0; i < B_IT; ++i) bufdma16(p.W, dB + (i * NLW + lw) * 1024, bvoff[i], bso, nr);
      };
      if (r1 > r0) issue_sc(r0, true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int r = r0; r < r1; ++r) {
        asm volatile("s_barrier" ::: "memory");           // S_r
        issue_sc(r + 1, r + 1 < r1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
#ifdef AGD_STAMPS
    if (ts_on) { AGD_TS(6); g_smap_ts[1023] = ts_n; g_smap_ts[1021] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if constexpr (!SPLITK) igemm_epilogue_ghost(p);
    return;
  }

  // =========================== consumer waves ===========================
  const int wm = wid / WN, wn = wid % WN;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, q = lane >> 4;
  int aaddr[3][MI];                                      // kk = 0 byte offset inside an A image; kk = 1 is ^ 64
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int pl = wm * WTM + i * 16 + frow;             // tile-local pixel
    const int j = pl / Wt;
    const int R0 = j * hw2 + (pl - j * Wt);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) { const int R = R0 + kx; aaddr[kx][i] = R * 128 + ((q ^ (R & 7)) << 4); }
  }
  int foffB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foffB[kk] = wn * WTN * 128 + (frow >> 2) * (4 * NI * 128) + (frow & 3) * 128 + ((((kk << 2) + q) ^ (lane & 7)) << 4);

  bf16x8 fa[2][MI], fb[2][NI];                           // [kk]: the two halves of a K step
  // fragments of half kk of the step (A image slot `as`, tap KX, weight stage slot `bs`)
  auto rd = [&](auto kx_tag, auto kk_tag, int as, int bs) {
    constexpr int KX = decltype(kx_tag)::value, kk = decltype(kk_tag)::value;
    const char* sA = sAr + as * A_BYTES;
    const char* sB = sBr + bs * B_BYTES;
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[kk][i] = *(const bf16x8*)(sA + (kk ? aaddr[KX][i] ^ 64 : aaddr[KX][i]));
#pragma unroll
    for (int j = 0; j < NI; ++j) fb[kk][j] = *(const bf16x8*)(sB + j * 512 + foffB[kk]);
  };
  auto mm = [&](auto kk_tag) {
    constexpr int kk = decltype(kk_tag)::value;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);      // D = W . X^T
  };
  // the half step's schedule: one fragment read behind each of the first nine MFMAs (a read between two MFMAs costs the matrix pipe nothing; nine in a row idle it
  // ~100 cycles), the other eleven MFMAs cover the last reads' latency before the next half step consumes them
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < MI + NI; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI - (MI + NI), 0);
  };
  // one K step: [B_t] read kk1(t) | MFMA kk0(t) | read kk0(t + 1) | MFMA kk1(t).  (as, bs): slots of step t; (asn, bsn): of step t + 1 (tap KXN)
  auto step = [&](auto kx_tag, auto kxn_tag, int as, int bs, int asn, int bsn) {
    __builtin_amdgcn_sched_barrier(0);
    AGD_TS(2);
    asm volatile("s_barrier" ::: "memory");
    AGD_TS(3);
    __builtin_amdgcn_sched_barrier(0);
    rd(kx_tag, K1{}, as, bs);
    mm(K0{});
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    rd(kxn_tag, K0{}, asn, bsn);                          // (past the last step: re-reads the last step's slots, unused)
    mm(K1{});
    interleave();
  };
  __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): no scalar load pending on the loop's entry edge (igemm_pc.h)
  asm volatile("s_barrier" ::: "memory");                // P: image g0 and stage 0 have landed
  if (nsteps > 0) rd(K0{}, K0{}, 0, 0);
  int as = 0, bs = 0;                                    // slots of the current group's image / the current step's weights
  for (int g = g0; g < g1; ++g) {
    const int asn = as + 1 == NA ? 0 : as + 1;
    const int b1 = bs + 1 >= NB ? bs + 1 - NB : bs + 1, b2 = bs + 2 >= NB ? bs + 2 - NB : bs + 2, b3 = bs + 3 >= NB ? bs + 3 - NB : bs + 3;
    step(K0{}, K1{}, as, bs, as, b1);
    step(K1{}, K2{}, as, b1, as, b2);
    step(K2{}, K0{}, as, b2, asn, b3);                    // (behind the last group: prefetches from slots nobody filled -- in bounds, never used)
    as = asn; bs = b3;
  }
  if (p.sc0) {                                           // the conv_shortcut chunks (loader side above): centre tap of image / weight slot (r - r0) & 1
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");              // Z
    const int nsc = (p.sc_C0 + p.sc_C1) >> 6;
    int r0 = 0, r1 = nsc;
    if constexpr (SPLITK) { const int per = (nsc + (int)gridDim.z - 1) / (int)gridDim.z; r0 = (int)blockIdx.z * per; r1 = r0 + per < nsc ? r0 + per : nsc; if (r1 < r0) r1 = r0; }
    for (int r = r0; r < r1; ++r) {
      const int slot = (r - r0) & 1;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");            // S_r
      __builtin_amdgcn_sched_barrier(0);
      rd(K1{}, K0{}, slot, slot);
      rd(K1{}, K1{}, slot, slot);
      __builtin_amdgcn_sched_barrier(0);
      mm(K0{});
      mm(K1{});
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  AGD_TS(5);
  igemm_epilogue<BM, BN, WM, WN, 0, SPLITK>(p, acc, smem, lane, wm, wn, m0, n0, tn, 0, nullptr);
#ifdef AGD_STAMPS
  if (ts_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); AGD_TS(6); g_smap_ts[1023] = ts_n; g_smap_ts[1021] = __builtin_amdgcn_s_memrealtime(); }
#endif
}
