// Weight-streaming 1x1 igemm for the small feature maps (round 4): W fragments straight to registers, activations through LDS.
//
// The 16 x 16 / 32 x 32 maps' linears (C -> C projections, qkv, ff.net.2) are INGEST bound in igemm_kernel:
// both operands enter the CU through the LDS-DMA path (~30 B/clk/CU), and at M = 2048, K = N = 1280 a 64 x 160 tile moves 573 KB through it
// for 26 MFLOP (16.7 us per launch against a 3 us MFMA floor).  Here the weight matrix -- the larger operand of these launches -- is
// re-laid at load time in MFMA FRAGMENT ORDER (tblock.hip, launch_frag_order_w), so that the 1 KiB a wave's A-operand needs is one
// coalesced 16-byte-per-lane global load: it goes straight to registers through the vector-memory path (64 B/clk/CU), in parallel with
// the activation tile, which alone still goes through LDS (register-staged: plain loads + ds_write, so that every load of the loop is
// of ONE kind and hipcc's counted waits stay counted -- cdna_hip_programming.md, 'Three .s-level traps' (b)).
//   tile 64 rows x (4 waves x NI x 16) columns, waves 1 x 4: every weight fragment is fetched by exactly one wave of the workgroup and
//   feeds four MFMAs (the four 16-row tiles); operand roles swapped as in igemm_kernel, so igemm_epilogue.h runs unchanged behind it.
#pragma once
#include "igemm_epilogue.h"

// weight-fragment ring (registers), filled WR_D fragments ahead: 8 / 6 for the plain kernels; the GEGLU form (NI = 4: eight fragments per stage) keeps two stages of
// fragments (16 / 14) in flight -- its 26 MB matrices arrive cold from HBM in situ
#define WR_F (NI == 4 ? 16 : 8)
#define WR_D (NI == 4 ? 14 : 6)
// WN = 5: 160-wide tiles (N = 1280 at M = 2048: exactly 256 workgroups); waves 0 - 3 of a group stage the activations.
// KG = 2: two K groups of WN waves per workgroup -- group g walks the 64-deep stages g, g + 2, ... with its own activation ring, weight-fragment ring and
// accumulators and hands its sums to group 0 through LDS before the epilogue (igemm_kernel's KG = 2, where the LDS-DMA path made it lose on these tiles;
// here the operands arrive over the vector-memory path, which one K step per CU does not fill).  K must hold an even number of stages (launcher).
template <int NI, int GEGLU, int WN = 4, int KG = 1>
__global__ __launch_bounds__(WN * 64 * KG, 2) void igemm_wreg_kernel(const IgemmP p) {
  constexpr int BM = 64, BN = WN * NI * 16, WTN = NI * 16, MI = 4, NTG = WN * 64;
  constexpr int STG = 3, A_BYTES = BM * 128;           // activation ring: 3 stages of [64 rows][64 k] bf16, 128-byte rows, chunk ^ (row & 7)
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTG);
  char* const smem = smem_all + kg * (STG * A_BYTES);
  const int tid = threadIdx.x % NTG, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tn, tm;
  tile_of(bid, tiles_m, tiles_n, p.wmajor, p.xb_m, p.xb_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = (p.K >> 6) / KG;                       // this group's stages of 64 k: global stage KG * s + kg

  // cold-weight warm-up (p.warm, as in igemm_kernel): the launch's first workgroups stream the fragment-ordered matrix through the caches once, 1 / nb each
  if (p.warm) {
    const int tot = gridDim.x, nb = tot < 512 ? tot : 512;
    if ((int)blockIdx.x < nb) {
      const long long pieces = ((long long)p.N * p.K * 2) >> 10;
      const long long p0 = pieces * blockIdx.x / nb, p1 = pieces * (blockIdx.x + 1) / nb;
      char* wl = smem_all + KG * STG * A_BYTES + (kg * WN + wid) * 1024;
      for (long long pc = p0 + kg * WN + wid; pc < p1; pc += WN * KG) bufdma16(p.Wfrag, wl, (unsigned)(pc * 1024 + lane * 16), 0u);
    }
  }

  // activation staging: thread t (of the group's first four waves) fetches chunks (row = t / 8 + 32 u, chunk t % 8), u = 0, 1, of every stage
  const int arow = (tid & 255) >> 3, ach = tid & 7;
  const bf16_t* ap[2]; bool aok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) { const int m = m0 + arow + 32 * u; aok[u] = m < p.M; ap[u] = p.src0 + (long long)(aok[u] ? m : 0) * p.K + ach * 8; }
  const int awoff[2] = {(arow) * 128 + ((ach ^ (arow & 7)) << 4), (arow + 32) * 128 + ((ach ^ (arow & 7)) << 4)};
  u32x4 areg[2][2];                                     // two stages in flight
  const bool stager = WN == 4 || wid < 4;               // wave-uniform
  auto a_load = [&](int slot, int s) {                  // s: the group's stage
    if (stager) {
#pragma unroll
      for (int u = 0; u < 2; ++u) areg[slot][u] = aok[u] ? *(const u32x4*)(ap[u] + (KG * s + kg) * 64) : u32x4{0, 0, 0, 0};
    }
  };
  auto a_store = [&](int slot, int stg) {
    if (stager) {
#pragma unroll
      for (int u = 0; u < 2; ++u) *(u32x4*)(smem + stg * A_BYTES + awoff[u]) = areg[slot][u];
    }
  };

  // weight stream of this wave: fragments (k-step of 32, tile j) in consumption order, (n0 / 16 / NI + wid) is the wave's column range;
  // the group's fragment n = s * FPS + r (stage s of the group, r < FPS) is fragment (KG * s + kg) * FPS + r of the matrix
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wfrag, 0, (unsigned)((long long)p.N * p.K * 2), 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  constexpr int FPS = 2 * NI;
  const int KS = p.K >> 5, NFR = nk * FPS;
  const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(((n0 / WTN + wid) * KS) * NI) * 1024u);
  u32x4 ring[WR_F];
  auto ldw = [&](int n) {
    const int fg = KG == 1 ? n : (KG * (n / FPS) + kg) * FPS + n % FPS;
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, wbase + (unsigned)fg * 1024u, 0));
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LayerNorm-fold consumer (as igemm_kernel): one thread per tile row sums the producer's partial sums while the prologue loads are in flight and leaves (mean, rstd)
  // in LDS behind the rings -- the epilogue then reads two floats per row instead of chasing `slots` loads per row
  float* const lnst = (float*)(smem_all + KG * STG * A_BYTES + KG * WN * 1024);
  if (p.ln_stats && threadIdx.x < BM) {
    const int m = m0 + (int)threadIdx.x;
    float S = 0.f, Q = 0.f;
    if (m < p.M) for (int k = 0; k < p.ln_slots; ++k) { const f32x2 v = *(const f32x2*)(p.ln_stats + ((long long)m * p.ln_slots + k) * 2); S += v[0]; Q += v[1]; }
    const float mu = S * p.ln_invC;
    float var = Q * p.ln_invC - mu * mu; var = var < 0.f ? 0.f : var;
    *(f32x2*)(lnst + threadIdx.x * 2) = f32x2{mu, rsqrtf(var + p.ln_eps)};
  }
  // prologue: activation stages 0, 1 requested, the ring's head in flight, stage 0 into LDS
  a_load(0, 0);
  if (nk > 1) a_load(1, 1);
#pragma unroll
  for (int f = 0; f < WR_D; ++f) ring[f] = ldw(f < NFR ? f : 0);
  a_store(0, 0);
  __syncthreads();

  const int frow = lane & 15, fq = lane >> 4;
  const int foff[2] = {frow * 128 + (((0 + fq) ^ (lane & 7)) << 4), frow * 128 + (((4 + fq) ^ (lane & 7)) << 4)};
  // per stage: {store stage s+1 (requested two stages ago) | request stage s+2 | 2 k-steps of (NI weight fragments, 4 activation fragments,
  // 4 NI MFMAs)} | barrier.  Fragment f = 2 NI s + NI kk + j sits in ring slot f % WR_F: the stage loop is unrolled by WR_F / gcd so that
  // slots are compile-time -- 2 NI fragments per stage and WR_F = 8 give a period of 2 (NI = 2) or 1 (NI = 4) stages.
  constexpr int PER = WR_F / FPS > 0 ? WR_F / FPS : 1;
  static_assert(WR_F % FPS == 0 || FPS % WR_F == 0, "ring period");
  auto stage = [&](int s, auto ph_tag) {
    constexpr int PH = decltype(ph_tag)::value;         // s % PER
    const int cur = s % STG;
    if (s + 1 < nk) a_store((s + 1) & 1, (s + 1) % STG);
    if (s + 2 < nk) a_load(s & 1, s + 2);
    const char* sA = smem + cur * A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 xf[MI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int fl = (PH * FPS + kk * NI + j);          // fragment index within the period
        const int f = s * FPS + kk * NI + j;
        const int tgt = f + WR_D;
        ring[(fl + WR_D) % WR_F] = ldw(tgt < NFR ? tgt : NFR - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (j == 0) {
#pragma unroll
          for (int i = 0; i < MI; ++i) xf[i] = *(const bf16x8*)(sA + i * 2048 + foff[kk]);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[fl % WR_F]), xf[i], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  };
  {
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1 % PER>;
    int s = 0;
    for (; s + PER <= nk; s += PER) {
      stage(s, P0{});
      if constexpr (PER > 1) stage(s + 1, P1{});
    }
    if constexpr (PER > 1) { if (s < nk) stage(s, P0{}); }
  }
  if constexpr (KG == 2) {
    // group 1's accumulators -> LDS (both activation rings are free: the last stage ended with a barrier) -> group 0 adds them and finishes the tile
    f32x4* xch = (f32x4*)smem_all + (wid * MI * NI) * 64 + lane;
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
  igemm_epilogue<BM, BN, 1, WN, GEGLU, 0>(p, acc, smem_all, lane, 0, wid, m0, n0, tn, 0, p.ln_stats ? lnst : nullptr);
}
