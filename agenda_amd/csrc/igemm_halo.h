// 3x3 / stride 1 / pad 1 implicit GEMM (optionally on the nearest-2x upsampled input) with a ROW-HALO A tile (included by igemm.hip).
//
// The general kernel streams one 128-row im2col tile per tap: nine LDS-DMA images of (almost) the same pixels per 64-channel chunk.
// Here the K loop runs (ky, source, channel chunk) groups; per group ONE A image holds the tile's pixel rows with one halo pixel on
// either side -- for a tile of `trows` image rows of Wt pixels: trows x (Wt + 2) LDS rows of 128 B -- and the three kx taps of the
// group read their fragments from it at row offsets 0 / 1 / 2 (fragment addresses are per lane, so the shift is free; the XOR swizzle
// key is the LDS row's, applied on the DMA source side as before).  A ingest per group: (Wt + 2) / (3 Wt) of the general kernel's;
// the weight ring (one tile per tap) is unchanged.  Two rings with different cadence: A slot = group parity, B slot = step parity.
#pragma once
#include "kernels.h"
#include "igemm_epilogue.h"
#include <type_traits>

template <int N> AGD_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int BN, int SPLITK, int BST>     // BST: weight ring depth, 2 (two workgroups per CU) or 4 (launches of <= 256 tiles: one per CU)
__global__ __launch_bounds__(256, BST == 2 ? 2 : 1) void igemm_halo_kernel(const IgemmP p, const int Wt, const int HRP) {
  constexpr int BM = 128, WM = 2, WN = 2, NT = 256, NW = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int B_IT = BN * 8 / NT, B_BYTES = BN * 128;
  constexpr int A_ITH = 5;                              // A image: up to 160 halo rows = 20 pieces of 8 rows, 5 per wave
  const int A_BYTES = HRP * 128;                        // HRP = the tile's halo rows, padded to 8; pieces past it land in a scratch KiB per wave
  static_assert((BN * 8) % NT == 0, "weight tile rows must split evenly over the DMA lanes");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sAr = smem;                               // A ring: 2 x A_BYTES
  char* const sBr = smem + 2 * A_BYTES;                 // B ring: BST x B_BYTES
  char* const scr = sBr + BST * B_BYTES;                // 4 KiB: dead-piece sink

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tn, tm;                                           // A-major walk (consecutive tiles share the pixel rows) or the launcher's XCD blocks
  tile_of(bid, (p.M + BM - 1) / BM, tiles_n, 0, p.xb_m, p.xb_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int W = p.Wout, H = p.Hout, HW = H * W;         // output grid (= the nearest-2x upsampled input grid when p.up == 2)
  const int ush = p.up == 2 ? 1 : 0;
  const int hw2 = Wt + 2;
  const int trows = BM / Wt;                            // image rows (or row segments) per tile
  const int Ct = p.C0 + p.C1;

  // ---- A gather state: piece i of this wave covers LDS halo rows (i*NW + wid)*8 .. +7; this lane fetches row lrow, 16-B chunk lchunk.
  // Rows past the tile's halo (and padding pixels) get an out-of-range offset: the hardware range check writes zeros.
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;                 // swizzle on the source side (LDS-DMA writes lane-linear)
  int a_brow[A_ITH], a_ix[A_ITH], a_y[A_ITH];           // image's first source row, source column, output row of tap ky = 0
  unsigned a_ok = 0;
#pragma unroll
  for (int i = 0; i < A_ITH; ++i) {
    const int R = (i * NW + wid) * 8 + lrow;
    const int j = R / hw2, col = R - j * hw2;
    const int pm = m0 + j * Wt;                         // first output pixel of tile row j
    bool ok = j < trows && pm < p.M;
    int b = 0, y = 0, x0 = 0;
    if (ok) { b = pm / HW; const int rem = pm - b * HW; y = rem / W; x0 = rem - y * W; }
    const int ix = x0 + col - 1;
    ok = ok && (unsigned)ix < (unsigned)W;
    a_brow[i] = b * p.Hin; a_ix[i] = ix >> ush;         // the upsample is folded into the gather: source pixel = (row >> 1, col >> 1)
    a_y[i] = y - 1;
    a_ok |= (ok ? 1u : 0u) << i;
  }
  unsigned bvoff[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = (i * NW + wid) * 8 + lrow;
    const int n = n0 + row;
    const int qp = (row % WTN) / (4 * NI);
    const int key = (row & 3) | ((qp & 1) << 2);        // permuted weight rows: see igemm.hip
    bvoff[i] = (n < p.N) ? (unsigned)(((long long)n * p.K + ((lane & 7) ^ key) * 8) * 2) : 0x80000000u;
  }

  // K range: groups (source, chunk, ky) -- the three ky images of a 64-channel chunk back to back: they are the same cache lines of the tile's pixel rows (one row up /
  // down), so A crosses L2 -> HBM once per launch instead of once per ky (FETCH_SIZE of the 16 x 16 maps' 1280 -> 1280 conv: 150 -> ~75 MB; an XCD's A panels exceed its
  // 4 MB L2 from C ~ 1000 on); split-K slices whole groups
  const int gpk = Ct >> 6;                              // chunks (groups per ky)
  const int ngr = 3 * gpk;
  int g0 = 0, g1 = ngr;
  if constexpr (SPLITK) {
    const int per = (ngr + (int)gridDim.z - 1) / (int)gridDim.z;
    g0 = (int)blockIdx.z * per; g1 = g0 + per < ngr ? g0 + per : ngr;
    if (g1 < g0) g1 = g0;
  }

  // per-group A source: offsets of the five pieces + descriptor base; recomputed once per group (VALU under the MFMAs).
  // (ky, r) of a group are carried incrementally (no division in the loop); weight offset of tap (ky, kx), chunk r:
  // ((ky*3 + kx) * Ct + r*64) * 2 bytes
  unsigned avoff[A_ITH]; unsigned aso = 0; const bf16_t* abase = p.src0;
  const int c0n = p.C0 >> 6;
  auto setA = [&](int ky, int r) {
    const bool s1 = r >= c0n;
    const int Cs = s1 ? p.C1 : p.C0;
    aso = (unsigned)((s1 ? r - c0n : r) * 128);
    abase = s1 ? p.src1 : p.src0;
#pragma unroll
    for (int i = 0; i < A_ITH; ++i) {
      const bool ok = ((a_ok >> i) & 1) && (unsigned)(a_y[i] + ky) < (unsigned)H;
      const int pix = (a_brow[i] + ((a_y[i] + ky) >> ush)) * p.Win + a_ix[i];
      avoff[i] = ok ? (unsigned)(pix * Cs + lchunk * 8) * 2u : 0x80000000u;                      // < 2^31 bytes per source (launcher)
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: A rows are per lane (pixel -> halo row, + kx per tap); B as in the general kernel
  const int frow = lane & 15, q = lane >> 4;
  int aaddr[3][MI];                                     // kk = 0 byte offset inside the A slot; kk = 1 is ^ 64
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int pl = wm * WTM + i * 16 + frow;            // tile-local pixel
    const int j = pl / Wt;
    const int R0 = j * hw2 + (pl - j * Wt);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) { const int R = R0 + kx; aaddr[kx][i] = R * 128 + ((q ^ (R & 7)) << 4); }
  }
  int foffB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int sw = (((kk << 2) + q) ^ (lane & 7)) << 4;
    foffB[kk] = (frow >> 2) * (4 * NI * 128) + (frow & 3) * 128 + sw;
  }

  // one K step (tap kx of group g): barrier, kk0 fragment reads | (2 MFMA, 1 LDS-DMA piece, 1 kk1 read) x n | remaining MFMAs.
  // Step kx = 0 also issues the NEXT group's A image (5 pieces, after the weight pieces: the next step waits with vmcnt(5), so
  // the image has two steps to land).  Dead pieces (past the K range) go through a zero-record descriptor: no branch in the body.
  // (kx_tag: the tap whose fragments are read; values >= 4 = tap (value - 4) AND the next A image issued in this step -- the 3x3 loop issues it in its kx = 0
  //  step (tag 4), the 1x1 shortcut loop in every step (tag 5: centre tap))
  auto kstep = [&](auto kx_tag, int aslot, int bslot, int bdst, unsigned bso, unsigned nrB, unsigned nrA) {
    constexpr int KXT = decltype(kx_tag)::value;
    constexpr int KX = KXT >= 4 ? KXT - 4 : KXT;
    constexpr bool ISSUE_A = KXT >= 4;
    constexpr int NF = MI + NI, ND = B_IT + (ISSUE_A ? A_ITH : 0), NG = NF > ND ? NF : ND;
    const char* sA = sAr + aslot * A_BYTES;
    const char* sB = sBr + bslot * B_BYTES + wn * WTN * 128;
    char* dB = sBr + bdst * B_BYTES;
    char* dA = sAr + (aslot ^ 1) * A_BYTES;
    const unsigned bso_u = __builtin_amdgcn_readfirstlane(bso), aso_u = __builtin_amdgcn_readfirstlane(aso);
    bf16x8 a0[MI], b0[NI], a1[MI], b1[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) a0[i] = *(const bf16x8*)(sA + aaddr[KX][i]);
#pragma unroll
    for (int j = 0; j < NI; ++j) b0[j] = *(const bf16x8*)(sB + j * 512 + foffB[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g < B_IT) bufdma16(p.W, dB + (g * NW + wid) * 1024, bvoff[g < B_IT ? g : 0], bso_u, nrB);
      else if (g < ND) { const int pc = (g - B_IT) * NW + wid;
        bufdma16(abase, pc * 8 < HRP ? dA + pc * 1024 : scr + wid * 1024, avoff[(g >= B_IT && g < ND) ? g - B_IT : 0], aso_u, nrA); }
      if (g < MI) a1[g] = *(const bf16x8*)(sA + (aaddr[KX][g < MI ? g : 0] ^ 64));
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
  static_assert(2 * (MI + NI > B_IT + A_ITH ? MI + NI : B_IT + A_ITH) <= 2 * MI * NI, "interleave needs enough MFMAs");

  using K0 = std::integral_constant<int, 4>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
  using KS1 = std::integral_constant<int, 5>;          // shortcut loop: centre tap, next A image issued
  constexpr unsigned LIVE = 0x7FFFFFF0u;
  if (p.warm == 2) {                                    // cold-weight warm-up (see igemm.hip): the first workgroups stream W once
    const int lin = blockIdx.z * gridDim.x + blockIdx.x;
    const int tot = gridDim.x * gridDim.z, nb = tot < 512 ? tot : 512;
    if (lin < nb) {
      const long long pieces = ((long long)p.N * p.K * 2) >> 10;
      const long long p0 = pieces * lin / nb, p1 = pieces * (lin + 1) / nb;
      for (long long pc = p0 + wid; pc < p1; pc += NW) bufdma16(p.W, scr + wid * 1024, (unsigned)(pc * 1024 + lane * 16), 0u);
    }
  }
#ifdef AGD_KY_OUTER                                     // A/B builds only (tools/ab_bench_libs.sh): the former order, ky outer
  int ky = g0 / gpk, r = g0 - ky * gpk;
#else
  int r = g0 / 3, ky = g0 - r * 3;                      // current group: chunk r, ky
#endif
  unsigned bso = (unsigned)(((ky * 3) * Ct + r * 64) * 2);   // weights of (ky, kx = 0, r)
  const unsigned tapb = (unsigned)(Ct * 2);            // one tap further
  if (g1 > g0) {                                        // prologue: A image of the first group, weights of its first tap(s)
    setA(ky, r);
    const unsigned aso_u = __builtin_amdgcn_readfirstlane(aso);
#pragma unroll
    for (int i = 0; i < A_ITH; ++i) { const int pc = i * NW + wid; bufdma16(abase, pc * 8 < HRP ? sAr + pc * 1024 : scr + wid * 1024, avoff[i], aso_u); }
#pragma unroll
    for (int t = 0; t < (BST == 2 ? 1 : 3); ++t) {
      const unsigned bso_u = __builtin_amdgcn_readfirstlane(bso + (unsigned)t * tapb);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) bufdma16(p.W, sBr + t * B_BYTES + (i * NW + wid) * 1024, bvoff[i], bso_u);
    }
  }
  // Waits (vmcnt is in order).  BST = 2: the step's weights were issued one step ago, after them only (in step kx = 0) the next A
  // image.  BST = 4: the weights were issued three steps ago; younger than them are two more weight tiles, plus the next A image
  // when it was issued in between.
  constexpr int W_K0 = BST == 2 ? 0 : 2 * B_IT, W_K1 = BST == 2 ? A_ITH : 2 * B_IT + A_ITH, W_K2 = BST == 2 ? 0 : 2 * B_IT + A_ITH;
  int bs = 0;                                           // ring slot of the current step's weights
  for (int g = g0; g < g1; ++g) {
    const int aslot = (g - g0) & 1;
    const bool more = g + 1 < g1;
#ifdef AGD_KY_OUTER
    int kyn = ky, rn = r + 1;
    if (rn == gpk) { rn = 0; ++kyn; }
#else
    int kyn = ky + 1, rn = r;                           // next group
    if (kyn == 3) { kyn = 0; ++rn; }
#endif
    const unsigned bson = (unsigned)(((kyn * 3) * Ct + rn * 64) * 2);
    if (more) setA(kyn, rn);                            // descriptors of the next group's image (issued inside step kx = 0)
    const unsigned nrn = more ? LIVE : 0u;
    if constexpr (BST == 2) {                           // weights of the NEXT step go out in this one
      wait_vm<W_K0>(); asm volatile("s_barrier" ::: "memory");
      kstep(K0{}, aslot, bs, bs ^ 1, bso + tapb, LIVE, nrn);
      wait_vm<W_K1>(); asm volatile("s_barrier" ::: "memory");
      kstep(K1{}, aslot, bs ^ 1, bs, bso + 2 * tapb, LIVE, 0u);
      wait_vm<W_K2>(); asm volatile("s_barrier" ::: "memory");
      kstep(K2{}, aslot, bs, bs ^ 1, bson, nrn, 0u);
      bs ^= 1;
    } else {                                            // weights of the same tap of the NEXT group (three steps ahead)
      wait_vm<W_K0>(); asm volatile("s_barrier" ::: "memory");
      kstep(K0{}, aslot, bs, (bs + 3) & 3, bson, nrn, nrn);
      wait_vm<W_K1>(); asm volatile("s_barrier" ::: "memory");
      kstep(K1{}, aslot, (bs + 1) & 3, bs, bson + tapb, nrn, 0u);
      wait_vm<W_K2>(); asm volatile("s_barrier" ::: "memory");
      kstep(K2{}, aslot, (bs + 2) & 3, (bs + 1) & 3, bson + 2 * tapb, nrn, 0u);
      bs = (bs + 3) & 3;
    }
    ky = kyn; r = rn; bso = bson;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // dead tail pieces still write zeros: let them land before the epilogue reuses LDS
  {
    if (p.sc0) {
      // ---- the block's 1x1 conv_shortcut, same accumulators: chunks of 64 channels of the raw block input (one or two sources), weight columns 9 Ct + 64 r ...;
      // two-slot rings (A images 0 / 1, weight slots 0 / 1), everything of a step requested one step ahead and waited for in full
      asm volatile("s_barrier" ::: "memory");          // every wave has left the 3x3 loop: the rings are free
      const int nsc = (p.sc_C0 + p.sc_C1) >> 6, sc0n = p.sc_C0 >> 6;
      int r0 = 0, r1 = nsc;                              // split-K: the chunks are dealt to the K slices like the 3x3 groups
      if constexpr (SPLITK) { const int per = (nsc + (int)gridDim.z - 1) / (int)gridDim.z; r0 = (int)blockIdx.z * per; r1 = r0 + per < nsc ? r0 + per : nsc; if (r1 < r0) r1 = r0; }
      auto setS = [&](int r) {                           // A image of shortcut chunk r: the tile's own pixel rows (ky = 1), source sc0 / sc1
        const bool s1 = r >= sc0n;
        const int Cs = s1 ? p.sc_C1 : p.sc_C0;
        aso = (unsigned)((s1 ? r - sc0n : r) * 128);
        abase = s1 ? p.sc1 : p.sc0;
#pragma unroll
        for (int i = 0; i < A_ITH; ++i) {
          const bool ok = ((a_ok >> i) & 1) && (unsigned)(a_y[i] + 1) < (unsigned)H;
          const int pix = (a_brow[i] + ((a_y[i] + 1) >> ush)) * p.Win + a_ix[i];
          avoff[i] = ok ? (unsigned)(pix * Cs + lchunk * 8) * 2u : 0x80000000u;
        }
      };
      const unsigned scb = (unsigned)(9 * Ct * 2);      // byte offset of the shortcut columns inside a weight row
      if (r1 > r0) {
        setS(r0);
        const unsigned aso_u = __builtin_amdgcn_readfirstlane(aso);
#pragma unroll
        for (int i = 0; i < A_ITH; ++i) { const int pc = i * NW + wid; bufdma16(abase, pc * 8 < HRP ? sAr + pc * 1024 : scr + wid * 1024, avoff[i], aso_u); }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) bufdma16(p.W, sBr + (i * NW + wid) * 1024, bvoff[i], __builtin_amdgcn_readfirstlane(scb + (unsigned)r0 * 128u));
      }
      for (int r = r0; r < r1; ++r) {
        const bool more = r + 1 < r1;
        const int sl = (r - r0) & 1;
        if (more) setS(r + 1);
        wait_vm<0>(); asm volatile("s_barrier" ::: "memory");
        kstep(KS1{}, sl, sl, sl ^ 1, scb + (unsigned)(r + 1) * 128u, more ? LIVE : 0u, more ? LIVE : 0u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  igemm_epilogue<BM, BN, WM, WN, 0, SPLITK>(p, acc, smem, lane, wm, wn, m0, n0, tn, 0);
}
