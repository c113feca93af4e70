// Producer / consumer implicit GEMM for the launches that are ONE workgroup per CU (the 16 x 16 and 8 x 8 maps of the UNet at batch 8):
// LOADER waves do nothing but issue the LDS-DMA pieces of the ring, CONSUMER waves nothing but fragment reads and MFMAs.
//
// Why (round 5, tools/ubench/l2_ingest.hip + DESIGN section 4): a CU takes in 50 - 63 B per clock from its XCD's L2 over the LDS-DMA path when waves
// issue pieces back to back (120 - 150 GB/s per CU, 4 - 8 waves) -- twice what the one-workgroup-per-CU launches of igemm_kernel get.  There a
// wave is loader AND consumer: its instruction stream is in order, an LDS-DMA piece holds it for ~80 cycles (the CU retires one 1-KiB piece per
// ~18 - 20 cycles, four waves take turns), and the 7 pieces of a 64 x 160 tile's K step cost the wave ~600 cycles around 160 cycles of its own
// MFMA issue: the K step lasts ~900 cycles for 320 cycles of matrix-pipe work whatever the ring depth ("costs the same with every load dropped":
// a dropped piece still pays its issue).  With the roles split over different waves of the SIMD the piece issue of the loaders and the
// fragment reads / MFMAs of the consumers overlap; the K step shrinks to the CU's piece rate (28 pieces x ~18 - 20 cycles for 64 x 160).
//
// Structure: 4 consumer waves (2 x 2 over the tile, igemm_kernel's fragment layout, swizzles, weight-row permutation and register epilogue) +
// NLW loader waves (each LPW = pieces / NLW LDS-DMA instructions per stage: equal counts, so one counted vmcnt per wave); STAGES-deep ring
// (5 x 28 KB for 64 x 160); ONE s_barrier per 64-deep K step for all waves -- loaders arrive when their share of the step's stage has landed,
// consumers when they have read the previous one -- after it the loaders refill the slot the consumers just left.  1x1 launches over one or
// two sources (channel concat), per-image weights, folded-LayerNorm consumers, row / column statistics producers: everything the epilogue of
// igemm_epilogue.h carries (the loaders keep its workgroup barriers company, igemm_epilogue_ghost).  KS = 3: stride-1 / pad-1 3x3 convs --
// the loaders compute the im2col offsets of every (tap, chunk) step themselves (they have the issue slots to spare), optional split-K.
#pragma once
#include "igemm_epilogue.h"

template <int BM, int BN, int NLW, int STAGES>
struct PcGeom {
  static constexpr int WM = 2, WN = 2, NW = 4;
  static constexpr int A_Q = BM / 8, B_Q = BN / 8, PQ = A_Q + B_Q;      // 1-KiB pieces (8 rows x 128 B) per stage
  static_assert(PQ % NLW == 0, "equal LDS-DMA counts per loader wave");
  static constexpr int LPW = PQ / NLW;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int THREADS = (NW + NLW) * 64;
  static constexpr int LDS = STAGES * STAGE + NLW * 1024 + BM * 8;       // ring + a sink KiB per loader wave + (mean, rstd) of the tile's rows
  static_assert(LDS <= 160 * 1024, "LDS");
  static_assert((STAGES - 2) * LPW < 64 && STAGES >= 4, "vmcnt field; the consumers' fragment prefetch needs one stage more");
};

template <int BM, int BN, int NLW, int STAGES, int KS, int GEGLU, int SPLITK>
__global__ __launch_bounds__((4 + NLW) * 64) void igemm_pc_kernel(const IgemmP p) {
  using G = PcGeom<BM, BN, NLW, STAGES>;
  constexpr int WM = G::WM, WN = G::WN, NW = G::NW, LPW = G::LPW, STAGE = G::STAGE, A_BYTES = G::A_BYTES;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  static_assert(KS == 1 || KS == 3, "1x1 or 3x3");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tn, tm;
  tile_of(bid, (p.M + BM - 1) / BM, tiles_n, p.wmajor, p.xb_m, p.xb_n, tm, tn);
#ifdef AGD_EXPERIMENTS
  if (p.dbg & 8) { tm = 0; tn = 0; }                  // timing variant (tools/kb_pc_parts.py): every workgroup computes tile (0, 0) -- one A panel and one W panel for the whole chip
#endif
  const int m0 = tm * BM, n0 = tn * BN;
  const int HWo = p.Hout * p.Wout;
  // K steps of this block: 1x1: the chunks of source 0 then source 1; 3x3: (tap, chunk) with the tap outer, as the weight rows are laid out
  const int nk0 = p.C0 >> 6, nkc = (p.C0 + p.C1) >> 6;
  const int nk_total = p.K >> 6;
  int ks0 = 0, nk = nk_total;
  if constexpr (SPLITK) {
    const int per = (nk_total + (int)gridDim.z - 1) / (int)gridDim.z;
    ks0 = (int)blockIdx.z * per;
    nk = nk_total - ks0 < per ? nk_total - ks0 : per;
    if (nk < 0) nk = 0;
  }
  float* const lnst = (float*)(smem + STAGES * STAGE + NLW * 1024);

  if (wid >= NW) {
    // =========================== loader waves ===========================
    const int lw = wid - NW;
    const bf16_t* base0 = p.src0;
    const bf16_t* base1 = p.src1 ? p.src1 : p.src0;
    const bf16_t* baseW = p.W + (p.w_per_image ? (long long)(m0 / HWo) * p.sW : 0);
    const int lrow = lane >> 3;
    // per piece of this wave: A piece (tile rows 8g ..) or B piece (weight rows 8(g - A_Q) ..)
    unsigned voff0[LPW], voff1[LPW];                // A: byte offset of this lane's 16 B in source 0 / 1 (1x1); B: in W
    int arow_b[LPW], arow_y[LPW], arow_x[LPW];     // KS = 3: the output pixel of this lane's A row
    bool a_ok[LPW];
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int g = lw + NLW * j;
      if (g < G::A_Q) {
        const int m = m0 + g * 8 + lrow;
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const unsigned ch = (unsigned)(((lane & 7) ^ lrow) * 16);
        if constexpr (KS == 1) {
          voff0[j] = a_ok[j] ? (unsigned)((long long)mm * p.C0 * 2) + ch : 0x80000000u;
          voff1[j] = a_ok[j] ? (unsigned)((long long)mm * p.C1 * 2) + ch : 0x80000000u;
          arow_b[j] = arow_y[j] = arow_x[j] = 0;
        } else {
          const int b = mm / HWo, rem = mm - b * HWo, oy = rem / p.Wout;
          arow_b[j] = b; arow_y[j] = oy - 1; arow_x[j] = rem - oy * p.Wout - 1;       // stride 1, pad 1
          voff0[j] = ch; voff1[j] = ch;
        }
      } else {
        const int row = (g - G::A_Q) * 8 + lrow, n = n0 + row;
        const int qp = (row % WTN) / (4 * NI);
        const int key = (row & 3) | ((qp & 1) << 2);      // the fragment row that reads this weight row (igemm_epilogue.h's permutation), mod 8
        voff0[j] = (n < p.N) ? (unsigned)(((long long)n * p.K + ((lane & 7) ^ key) * 8) * 2) : 0x80000000u;
        voff1[j] = voff0[j];
        a_ok[j] = true; arow_b[j] = arow_y[j] = arow_x[j] = 0;
      }
    }
    char* const sink = smem + STAGES * STAGE + lw * 1024;
    // cold-weight warm-up (as igemm_kernel): A-major launches with a large matrix -- the first workgroups stream W once, 1 / nb each, through the caches
    if (p.warm == 2) {
      const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const int tot = gridDim.x * gridDim.y * gridDim.z, nb = tot < 512 ? tot : 512;
      if (lin < nb) {
        const long long pieces = ((long long)p.N * p.K * 2) >> 10;
        const long long p0 = pieces * lin / nb, p1 = pieces * (lin + 1) / nb;
        for (long long pc = p0 + lw; pc < p1; pc += NLW) bufdma16(baseW, sink, (unsigned)(pc * 1024 + lane * 16), 0u);
      }
    } else if (p.warm == 1 && blockIdx.z == 0 && (blockIdx.x >> 3) < 64) {
      // W-major launches: every XCD's first workgroups stream the contiguous slice of W their XCD's tiles will read
      const int nwg = gridDim.x, x = blockIdx.x & 7, jx = blockIdx.x >> 3;
      const int q8 = nwg >> 3, r8 = nwg & 7;
      const int lbase = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8, lcnt = q8 + (x < r8 ? 1 : 0);
      if (lcnt > 0 && p.wmajor) {
        const int tiles_m = (p.M + BM - 1) / BM;
        const int tn0 = lbase / tiles_m, tn1 = (lbase + lcnt - 1) / tiles_m;
        const int r0 = tn0 * BN, r1 = (tn1 + 1) * BN < p.N ? (tn1 + 1) * BN : p.N;
        const long long b0 = (long long)r0 * p.K * 2, pieces = ((long long)(r1 - r0) * p.K * 2) >> 10;
        const long long p0 = pieces * jx / 64, p1 = pieces * (jx + 1) / 64;
        for (long long pc = p0 + lw; pc < p1; pc += NLW) bufdma16(baseW, sink, (unsigned)(b0 + pc * 1024 + lane * 16), 0u);
      }
    }
    // one stage = LPW pieces of this wave; steps are issued in order (a cursor walks the block's K range: 1x1 = chunks of source 0 then source 1,
    // 3x3 = (tap, chunk) with k = tap * (C0 + C1) + c); steps past the range go through zero-record descriptors (dropped; the hardware writes zeros)
    int cur_s = 0, cur_cc = ks0, cur_tap = 0;
    if constexpr (KS == 3) { cur_tap = ks0 / nkc; cur_cc = ks0 - cur_tap * nkc; }
#ifdef AGD_EXPERIMENTS
    const bool nodma = (p.dbg & 1) != 0;                      // timing variant (tools/kb_pc_parts.py): no LDS-DMA instructions at all (results are garbage)
#else
    constexpr bool nodma = false;
#endif
    auto issue = [&](int slot) {
      const bool live = cur_s < nk;
      char* const sA = ring + slot * STAGE;
      char* const sB = sA + A_BYTES;
      const unsigned nr = live ? 0x7FFFFFF0u : 0u;
      const unsigned bso = __builtin_amdgcn_readfirstlane((unsigned)(ks0 + (live ? cur_s : 0)) * 128u);
      const bool s1 = __builtin_amdgcn_readfirstlane((int)(cur_cc >= nk0)) != 0;
      const bf16_t* bA = s1 ? base1 : base0;
      const unsigned aso = __builtin_amdgcn_readfirstlane((unsigned)(s1 ? cur_cc - nk0 : cur_cc) * 128u);
      if constexpr (KS == 1) {
#pragma unroll
        for (int j = 0; j < LPW; ++j) {
          const int g = lw + NLW * j;
          if (nodma) continue;
          if (g < G::A_Q) bufdma16(bA, sA + g * 1024, s1 ? voff1[j] : voff0[j], aso, nr);
          else bufdma16(baseW, sB + (g - G::A_Q) * 1024, voff0[j], bso, nr);
        }
      } else {
        const int kh = cur_tap / 3, kw = cur_tap - kh * 3;
        const int Cs = s1 ? p.C1 : p.C0;
#pragma unroll
        for (int j = 0; j < LPW; ++j) {
          const int g = lw + NLW * j;
          if (nodma) continue;
          if (g < G::A_Q) {
            const int iy = arow_y[j] + kh, ix = arow_x[j] + kw;
            const bool ok = a_ok[j] && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const unsigned off = ok ? (unsigned)((long long)((arow_b[j] * p.Hin + iy) * p.Win + ix) * Cs * 2) + voff0[j] : 0x80000000u;
            bufdma16(bA, sA + g * 1024, off, aso, nr);
          } else bufdma16(baseW, sB + (g - G::A_Q) * 1024, voff0[j], bso, nr);
        }
      }
      if (live) { ++cur_s; if (++cur_cc == nkc) { cur_cc = 0; ++cur_tap; } }
    };
    // Protocol (the consumers read the fragments of step ks + 1 UNDER the MFMAs of step ks, so a stage must have landed one barrier earlier than it is
    // multiplied): barrier P -- stage 0 landed; barrier B_ks (one per K step) -- stage ks + 1 landed, and every consumer has the fragments of stage
    // ks - 1 in registers (it issued that step's MFMAs), so the loaders refill that slot with stage ks + STAGES - 1.
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue(s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"((STAGES - 2) * LPW) : "memory");        // this wave's pieces of stage 0 have landed
    asm volatile("s_barrier" ::: "memory");                                           // P
    for (int ks = 0; ks < nk; ++ks) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"((STAGES - 3) * LPW) : "memory");      // ... of stage ks + 1 (stages up to ks + STAGES - 2 are issued)
      asm volatile("s_barrier" ::: "memory");                                         // B_ks
      issue((ks + STAGES - 1) % STAGES);                       // stage ks + STAGES - 1 into the slot of stage ks - 1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (dead tail pieces still write zeros into LDS: let them land before the epilogue reuses it)
    if constexpr (!SPLITK) igemm_epilogue_ghost(p);
    return;
  }

  // =========================== consumer waves ===========================
  const int wm = wid / WN, wn = wid % WN;
  // folded LayerNorm: (mean, rstd) of the tile's rows, one thread per row, while the loaders fill the ring
  if (!SPLITK && p.ln_stats && threadIdx.x < BM) {
    const int m = m0 + (int)threadIdx.x;
    float S = 0.f, Q = 0.f;
    if (m < p.M) for (int k = 0; k < p.ln_slots; ++k) { const f32x2 v = *(const f32x2*)(p.ln_stats + ((long long)m * p.ln_slots + k) * 2); S += v[0]; Q += v[1]; }
    const float mu = S * p.ln_invC;
    float var = Q * p.ln_invC - mu * mu; var = var < 0.f ? 0.f : var;
    *(f32x2*)(lnst + threadIdx.x * 2) = f32x2{mu, rsqrtf(var + p.ln_eps)};
  }
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  int foff[2], foffB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int sw = (((kk << 2) + (lane >> 4)) ^ (lane & 7)) << 4;
    foff[kk] = frow * 128 + sw;
    foffB[kk] = (frow >> 2) * (4 * NI * 128) + (frow & 3) * 128 + sw;
  }
#ifdef AGD_EXPERIMENTS
  if (p.dbg & 6) {
    // timing variants (tools/kb_pc_parts.py; results are garbage): bit 1 = the consumers read no fragments (MFMAs on register constants), bit 2 = fragment reads but no MFMAs
    bf16x8 ca = {}, cb = {};
    ca[0] = (__bf16)(float)lane; cb[1] = (__bf16)1.0f;
    asm volatile("s_barrier" ::: "memory");                    // P
    for (int ks = 0; ks < nk; ++ks) {
      asm volatile("s_barrier" ::: "memory");
      const char* sA = ring + (ks % STAGES) * STAGE + wm * WTM * 128;
      const char* sB = ring + (ks % STAGES) * STAGE + A_BYTES + wn * WTN * 128;
      if (p.dbg & 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb, ca, acc[i][j], 0, 0, 0);
      } else {
        float t = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < MI; ++i) { const bf16x8 v = *(const bf16x8*)(sA + i * 2048 + foff[kk]); t += (float)v[0]; }
#pragma unroll
          for (int j = 0; j < NI; ++j) { const bf16x8 v = *(const bf16x8*)(sB + j * 512 + foffB[kk]); t += (float)v[0]; }
        }
        acc[0][0][0] += t;
      }
    }
  } else
#endif
  {
    // two fragment sets: the reads of step ks + 1 are issued right behind barrier B_ks and land under the MFMAs of step ks (one wave per SIMD has no
    // partner to hide the fragment-read latency behind: without this a K step was barrier -> reads -> wait -> MFMAs, ~0.4 us for 0.17 us of matrix-pipe work)
    bf16x8 fa[2][2][MI], fb[2][2][NI];
    auto rd = [&](bf16x8 (&ra)[2][MI], bf16x8 (&rb)[2][NI], int stage) {
      const char* sA = ring + (stage % STAGES) * STAGE + wm * WTM * 128;
      const char* sB = ring + (stage % STAGES) * STAGE + A_BYTES + wn * WTN * 128;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) ra[kk][i] = *(const bf16x8*)(sA + i * 2048 + foff[kk]);
#pragma unroll
        for (int j = 0; j < NI; ++j) rb[kk][j] = *(const bf16x8*)(sB + j * 512 + foffB[kk]);
      }
    };
    auto mm = [&](const bf16x8 (&ra)[2][MI], const bf16x8 (&rb)[2][NI]) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[kk][j], ra[kk][i], acc[i][j], 0, 0, 0);      // D = W . X^T
    };
    // the step's schedule: one fragment read of the NEXT step behind each of this step's first MFMAs (igemm_pch.h: a read between two MFMAs costs the matrix pipe
    // nothing, a burst of them idles it), the remaining MFMAs cover the last reads' latency
    auto interleave = [&]() {
      constexpr int NR = 2 * (MI + NI), NM = 2 * MI * NI;
      static_assert(NM >= NR, "more MFMAs than fragment reads per step");
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
    };
    // (sched_barrier(0): hipcc otherwise hoists MFMAs over the s_barrier)
    // (every scalar load -- kernel arguments -- is retired before the loop: with one pending on the loop's entry edge hipcc's wait-count pass can only emit
    //  lgkmcnt(0) inside it, scalar loads returning out of order, and the first MFMA would wait for the reads just issued instead of the previous step's)
    __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0)
    asm volatile("s_barrier" ::: "memory");                    // P: stage 0 has landed
    if (nk > 0) rd(fa[0], fb[0], 0);
    int ks = 0;
    for (; ks + 1 < nk; ks += 2) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");                  // B_ks: stage ks + 1 has landed
      __builtin_amdgcn_sched_barrier(0);
      rd(fa[1], fb[1], ks + 1);
      mm(fa[0], fb[0]);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");                  // B_(ks + 1)
      __builtin_amdgcn_sched_barrier(0);
      rd(fa[0], fb[0], ks + 2 < nk ? ks + 2 : nk - 1);         // (unconditional: a constant number of reads in flight lets hipcc emit counted lgkmcnt waits; past the end it re-reads the last stage, unused)
      mm(fa[1], fb[1]);
      interleave();
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks < nk) {                                             // odd step count: the last step's fragments are in set 0
      asm volatile("s_barrier" ::: "memory");
      mm(fa[0], fb[0]);
    }
  }
  igemm_epilogue<BM, BN, WM, WN, GEGLU, SPLITK>(p, acc, smem, lane, wm, wn, m0, n0, tn, 0, (!SPLITK && p.ln_stats && nk_total > 0) ? lnst : nullptr);
}
