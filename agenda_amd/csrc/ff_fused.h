// Fused GEGLU feed-forward of a BasicTransformerBlock (included by igemm.hip):
//   h_out = h + W2 . ( val(LN(h)) * gelu(gate(LN(h))) ) + b2          (ff.net.0.proj -> GEGLU -> ff.net.2)
// for blocks whose channel count lets the 128-row activation panel live in LDS (C = 320: the 64x64 layers).  The 4C-wide hidden
// activation never touches HBM.
//
// One workgroup = 128 rows, 4 waves x 32 rows.  A wave keeps its 32 rows of the panel in REGISTERS as MFMA fragments (80 VGPRs), which leaves
// the LDS to a 4-deep ring for GEMM1's weight tiles and a double-buffered GEMM2 weight tile (one wave per SIMD: latency must be hidden
// by distance, not by a second wave).
// Hidden channels go by in chunks of 64 = one GEGLU weight tile of 128 rows ([8 values | 8 gates] groups, the layout launch_convert_weight
// already produces):
//   GEMM1  S[128 cols][32 px] = W1_chunk . X^T over K = C (weight tiles of 64 k stream through a 2-stage ring);
//   the LayerNorm-folded GEGLU epilogue runs in registers: with the permuted weight rows a lane owns 32 consecutive columns of one
//   pixel = two [8|8] groups = 16 hidden channels -- as bf16x8 pairs these ARE the pixel-side fragments of a 16x16x32 MFMA whose k
//   order is (k-group q, step kk) <-> hidden q*16 + kk*8 + e.  W2's columns are stored in that order per chunk (`w2p`, built once);
//   GEMM2  O[C][32 px] += W2p_chunk . H^T straight from those registers (the chunk's W2 tile [C][64] arrives under GEMM1).
// Epilogue: + b2 + residual (the panel's own rows), bf16, 16-B stores -- a lane owns C/4 consecutive channels of its pixel.
#pragma once
#include "kernels.h"


template <int N> AGD_DEV void ff_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int C>
__global__ __launch_bounds__(256, 1) void ff_fused_kernel(const FfP p) {
  constexpr int KS1 = C / 64, HID = 4 * C, NCH = HID / 64;
  constexpr int MI = 2, NI1 = 8, NI2 = C / 16, LC = C / 4;       // LC: consecutive output channels per lane
  constexpr int W1_STAGE = 16384, W1_ST = 3, W2_BYTES = C * 128;
  constexpr int W2_IT = C / 32;                                   // W2 pieces (8 rows each) per wave and chunk
  constexpr int W2_PS = W2_IT / KS1;                              // ... per K step
  static_assert(C % 64 == 0 && NI2 % 2 == 0 && W2_IT % KS1 == 0 && W2_IT - 2 * W2_PS >= 0, "panel geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sW1 = smem;                                         // weight ring of GEMM1: W1_ST tiles of [128 rows][64 k]
  char* const sW2 = sW1 + W1_ST * W1_STAGE;                       // two chunk tiles [C rows][64 hidden] of GEMM2
  float* const sK = (float*)(sW2 + 2 * W2_BYTES);                 // GEGLU constants: colsum values | colsum gates | bias values | bias gates, HID each

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 128;
  const int lrow = lane >> 3;
  const int frow = lane & 15, q = lane >> 4;

  // ---- the activation panel lives in REGISTERS: this lane's MFMA fragments of its two pixel tiles over the whole K = C
  bf16x8 afr[KS1][2][MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wid * 32 + i * 16 + frow;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x4 v = u32x4{0, 0, 0, 0};
        if (m < p.M) v = *(const u32x4*)(p.h + (long long)m * C + ks * 64 + kk * 32 + q * 8);
        afr[ks][kk][i] = __builtin_bit_cast(bf16x8, v);
      }
  }

  // ---- DMA offsets
  unsigned b1v[4], w2v[W2_IT];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wid) * 8 + lrow;
    const int key = (row & 3) | (((row >> 5) & 1) << 2);           // W1 tile rows are read in permuted order: q' = row / 32
    b1v[i] = (unsigned)((row * C + ((lane & 7) ^ key) * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < W2_IT; ++i) {
    const int row = (i * 4 + wid) * 8 + lrow;                       // output channel
    const int key = (row & 3) | (((row / LC) & 1) << 2);
    w2v[i] = (unsigned)((row * HID + ((lane & 7) ^ key) * 8) * 2);
  }
  constexpr unsigned LIVE = 0x7FFFFFF0u;
  auto issue_w1 = [&](int tt, unsigned nr) {                        // W1 tile of global K step tt = chunk * KS1 + ks
    const int c = tt / KS1, ks = tt - c * KS1;
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((c * 128 * C + ks * 64) * 2));
#pragma unroll
    for (int i = 0; i < 4; ++i) bufdma16(p.w1, sW1 + (tt % W1_ST) * W1_STAGE + (i * 4 + wid) * 1024, b1v[i], so, nr);
  };
  auto issue_w2 = [&](int c, int first, int n, unsigned nr) {       // pieces [first, first + n) of chunk c's W2 tile
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(c * 128));
#pragma unroll
    for (int u = 0; u < W2_IT; ++u) if (u >= first && u < first + n) bufdma16(p.w2p, sW2 + (c & 1) * W2_BYTES + (u * 4 + wid) * 1024, w2v[u], so, nr);
  };
  // GEGLU constants into LDS (read back per chunk through the LDS counter, not vmcnt)
  for (int i = tid; i < HID; i += 256) { sK[i] = p.ln_cs[i]; sK[HID + i] = p.ln_cs[HID + i]; sK[2 * HID + i] = p.bias1[i]; sK[3 * HID + i] = p.bias1[HID + i]; }
  __syncthreads();
  // prologue in the steady-state order (4 W1 pieces then W2_PS W2 pieces per step), so that ONE wait count serves every step
  issue_w2(0, 0, W2_IT - 2 * W2_PS, LIVE);
#pragma unroll
  for (int s0 = 0; s0 < 2; ++s0) { issue_w1(s0, LIVE); issue_w2(0, W2_IT - 2 * W2_PS + s0 * W2_PS, W2_PS, LIVE); }
  constexpr int WAITN = (4 + W2_PS) + W2_PS;                        // younger than the W1 tile a step needs: its own step's W2 pieces + one step

  // ---- LayerNorm statistics of this lane's two rows
  float lmu[MI], lrs[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wid * 32 + i * 16 + frow;
    float S = 0.f, Q = 0.f;
    if (m < p.M) { const float* sp = p.ln_stats + (long long)m * p.ln_slots * 2; for (int k = 0; k < p.ln_slots; ++k) { S += sp[2 * k]; Q += sp[2 * k + 1]; } }
    const float mu = S * p.ln_invC;
    float var = Q * p.ln_invC - mu * mu; var = var < 0.f ? 0.f : var;
    lmu[i] = mu; lrs[i] = rsqrtf(var + p.ln_eps);
  }

  // ---- fragment offsets
  int foffB1[2], foffW2[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int sw = (((kk << 2) + q) ^ (lane & 7)) << 4;
    foffB1[kk] = (frow >> 2) * (4 * NI1 * 128) + (frow & 3) * 128 + sw;
    foffW2[kk] = (frow >> 2) * (LC * 128) + (frow & 3) * 128 + sw;
  }

  f32x4 acc2[MI][NI2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int TOT = NCH * KS1;
  int t = 0;                                                         // global K step
  for (int c = 0; c < NCH; ++c) {
    f32x4 acc1[MI][NI1];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks, ++t) {
      ff_wait_vm<WAITN>();
      asm volatile("s_barrier" ::: "memory");
      const char* sb = sW1 + (t % W1_ST) * W1_STAGE;
      // weight fragments eight ahead, then (1 fragment read, 2 MFMA) with the step's LDS-DMA pieces spread over the first MFMAs
      constexpr int ND = 4 + W2_PS;
      const int tn = t + 2;                                          // the W1 tile two steps ahead; the NEXT chunk's W2 tile in instalments
      const int cn = tn / KS1, ksn = tn - cn * KS1;
      const unsigned so1 = __builtin_amdgcn_readfirstlane((unsigned)((cn * 128 * C + ksn * 64) * 2));
      const unsigned so2 = __builtin_amdgcn_readfirstlane((unsigned)((c + 1) * 128));
      const unsigned nr1 = tn < TOT ? LIVE : 0u, nr2 = c + 1 < NCH ? LIVE : 0u;
      char* d1 = sW1 + (tn % W1_ST) * W1_STAGE;
      char* d2 = sW2 + ((c + 1) & 1) * W2_BYTES;
#pragma unroll
      for (int x = 0; x < 2 * NI1; ++x) {
#if defined(AGD_EXPERIMENTS) && defined(EXP_FF_NODMA)      // timing only: no LDS-DMA in the loop (garbage results)
        if (false) {} else if (false) {
#else
        if (x < 4) bufdma16(p.w1, d1 + (x * 4 + wid) * 1024, b1v[x < 4 ? x : 0], so1, nr1);
        else if (x < ND) {
#endif
          const int u = ks * W2_PS + (x - 4); bufdma16(p.w2p, d2 + (u * 4 + wid) * 1024, w2v[(x >= 4 && x < ND) ? u : 0], so2, nr2); }
        const bf16x8 w = *(const bf16x8*)(sb + (x % NI1) * 512 + foffB1[x / NI1]);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc1[i][x % NI1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, afr[ks][x / NI1][i], acc1[i][x % NI1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);            // eight weight fragments ahead (LDS latency ~ 8 MFMA pairs)
#pragma unroll
      for (int x = 0; x < 2 * NI1; ++x) {
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        if (x < ND) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        if (x < 2 * NI1 - 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    // ---- GEGLU (LayerNorm folded) in registers -> the pixel-side fragments of GEMM2
    float csv[2][8], csg[2][8], hbv[2][8], hbg[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      const int hid0 = c * 64 + q * 16 + g2 * 8;
#pragma unroll
      for (int v4 = 0; v4 < 2; ++v4) {
        *(f32x4*)&csv[g2][4 * v4] = *(const f32x4*)(sK + hid0 + 4 * v4);
        *(f32x4*)&csg[g2][4 * v4] = *(const f32x4*)(sK + HID + hid0 + 4 * v4);
        *(f32x4*)&hbv[g2][4 * v4] = *(const f32x4*)(sK + 2 * HID + hid0 + 4 * v4);
        *(f32x4*)&hbg[g2][4 * v4] = *(const f32x4*)(sK + 3 * HID + hid0 + 4 * v4);
      }
    }
    bf16x8 hf[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = acc1[i][4 * g2 + (e >> 2)][e & 3], xg = acc1[i][4 * g2 + 2 + (e >> 2)][e & 3];
#if defined(AGD_EXPERIMENTS) && defined(EXP_FF_NOGELU)     // timing only: no gelu
          v[e] = (lrs[i] * (xv - lmu[i] * csv[g2][e]) + hbv[g2][e]) * (lrs[i] * (xg - lmu[i] * csg[g2][e]) + hbg[g2][e]);
#else
          v[e] = (lrs[i] * (xv - lmu[i] * csv[g2][e]) + hbv[g2][e]) * gelu_erf_f(lrs[i] * (xg - lmu[i] * csg[g2][e]) + hbg[g2][e]);
#endif
        }
        u32x4 pk;
        pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]); pk[2] = pack_bf2(v[4], v[5]); pk[3] = pack_bf2(v[6], v[7]);
        hf[i][g2] = __builtin_bit_cast(bf16x8, pk);
      }
    // GEMM2: this chunk's W2 tile was issued during the PREVIOUS chunk (the prologue for chunk 0): every one of its pieces is older
    // than the WAITN youngest operations of this chunk's later steps, whose barriers published it.
    {
      const char* s2 = sW2 + (c & 1) * W2_BYTES;
#pragma unroll
      for (int x = 0; x < 2 * NI2; ++x) {
        const bf16x8 w = *(const bf16x8*)(s2 + (x % NI2) * 512 + foffW2[x / NI2]);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc2[i][x % NI2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, hf[i][x / NI2], acc2[i][x % NI2], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);             // eight fragments ahead, then one read per two MFMAs
#pragma unroll
      for (int x = 0; x < 2 * NI2 - 8; ++x) { __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x8, 16, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: + b2 + residual, bf16, 16-B chunks of this lane's LC consecutive channels
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wid * 32 + i * 16 + frow;
    if (m >= p.M) continue;
    const bf16_t* rp = p.h + (long long)m * C + q * LC;
    bf16_t* op = p.out + (long long)m * C + q * LC;
#pragma unroll
    for (int c8 = 0; c8 < LC / 8; ++c8) {
      const u32x4 r = *(const u32x4*)(rp + 8 * c8);
      float bb[8];
      *(f32x4*)&bb[0] = *(const f32x4*)(p.bias2 + q * LC + 8 * c8); *(f32x4*)&bb[4] = *(const f32x4*)(p.bias2 + q * LC + 8 * c8 + 4);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float res = (e & 1) ? __uint_as_float(r[e >> 1] & 0xFFFF0000u) : __uint_as_float(r[e >> 1] << 16);
        v[e] = acc2[i][2 * c8 + (e >> 2)][e & 3] + bb[e] + res;
      }
      u32x4 pk;
      pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]); pk[2] = pack_bf2(v[4], v[5]); pk[3] = pack_bf2(v[6], v[7]);
      *(u32x4*)(op + 8 * c8) = pk;
    }
  }
}

// w2p[row][c*64 + kk*32 + qg*8 + e] = w2[row][c*64 + qg*16 + kk*8 + e]
__global__ void ff_permute_w2_kernel(const bf16_t* __restrict__ w2, bf16_t* __restrict__ w2p, int rows, int hid) {
  const long long n = (long long)rows * hid;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % hid); const long long row = i / hid;
    const int c = col >> 6, pos = col & 63, kk = pos >> 5, qg = (pos >> 3) & 3, e = pos & 7;
    w2p[i] = w2[row * hid + c * 64 + qg * 16 + kk * 8 + e];
  }
}
